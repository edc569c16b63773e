#!/bin/bash
# Per-workload PMC counters of the current kernels (r4 #5: + the HBM-bound regime): C2, C3 shadow rays (any_hit), C4, random geometry
# (1 M tris), the 1 Mi-ray C3 view and 4 M incoherent rays on a 4 M-triangle BLAS, each in its own process and one counter set per pass (--pmc with --kernel-trace only; FETCH_SIZE / WRITE_SIZE in
# separate passes), plus one un-profiled --kernel-trace pass for the launch duration.  The last three dispatches of the workload's kernel
# are averaged: steady state, claim order learned from the previous launch of the same batch.
#   tools/pmc_workloads.sh r06      ->  gpurun_out/r06/profiles/r06_pmc_workloads_kernel5.json   (copy to profiles/)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${1:-r06}
O=gpurun_out/$R/wl; P=gpurun_out/$R/profiles
mkdir -p $O $P
for W in c2 shadow c4 r1m c3 hbm hbm16; do
  EXTRA=""; [ "$W" = "c3" ] && EXTRA="--c3res 1024"
  export RC_PROBE_REPS=7   # launches 5-7 of the batch are averaged: the learned order in use, none of them a recording launch (those are launches 2-4 and every 8th)
  CMD="python3 tools/perf_probe.py --variants kernel=-1 --workloads $W $EXTRA"
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/$W/trace -- $CMD > $O/$W.trace.log 2>&1
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TA_TA_BUSY_sum TD_TD_BUSY_sum" "GRBM_GUI_ACTIVE GRBM_COUNT" \
             "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
             "SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$W/pmc_$i -- $CMD > $O/$W.pmc_$i.log 2>&1
  done
done
python3 - "$R" <<'PY'
import csv, glob, json, sys, collections
R = sys.argv[1]
O, P = f"gpurun_out/{R}/wl", f"gpurun_out/{R}/profiles"
sys.path.insert(0, ".")
import bench
names = {"c2": ("BASELINE C2: 100 000-triangle BLAS, 1 000 000 coherent grid rays, closest_hit", 1000000, "<false"),
         "shadow": ("C3 scene, 2 077 338 any_hit shadow rays from the primary hit points toward the point light", 2077338, "<true"),
         "c4": ("BASELINE C4: C3 scene, 16 777 216 incoherent cosine-hemisphere bounce rays, closest_hit", 16777216, "<false"),
         "r1m": ("random geometry, 1 000 000 triangles in one BLAS, 1 000 000 coherent grid rays, closest_hit (the reference's published benchmark shape)", 1000000, "<false"),
         "c3": ("C3 scene, 1 048 576 primary rays (1024 x 1024 pinhole), closest_hit: the mid-size batch of the headline scene", 1048576, "<false"),
         "hbm": ("the HBM-bound regime: one 4 000 000-triangle BLAS of random triangles (512 MB of nodes: beyond L2 + Infinity Cache), 4 194 304 incoherent rays "
                 "(uniform origins in the unit cube, uniform directions, seed 7: the batch of bench.py's extra), closest_hit", 4194304, "<false"),
         "hbm16": ("past the Infinity Cache: one 16 000 000-triangle BLAS of random triangles (2 GB of nodes, 8 x the 256 MiB MALL), the same 4 194 304 incoherent rays, closest_hit", 4194304, "<false")}
out = {"command": "rocprofv3 --pmc <set> --kernel-trace --output-format csv -- python3 tools/perf_probe.py --variants kernel=-1 --workloads <w> (one process per workload and counter set; tools/pmc_workloads.sh)",
       "averaged": "the last 3 dispatches of the workload's trace kernel in each pass = launches 5-7 of the same batch: the learned claim order in use, no recording (a batch records its launches 2-4 and then one in 8, which run ~7 % longer)",
       "fingerprint": bench.kernel_fingerprint(), "workloads": {}}
for w, (desc, n_rays, mode_tag) in names.items():
    agg, meta = collections.defaultdict(list), {}
    for f in glob.glob(f"{O}/{w}/pmc_*/**/*_counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "k_trace_phased" in r["Kernel_Name"] and mode_tag in r["Kernel_Name"]]
        per = collections.defaultdict(dict)
        for r in rows:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
            meta = {k: r[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Scratch_Size") if k in r}
        for d in sorted(per)[-3:]:
            for k, v in per[d].items():
                agg[k].append(v)
    entry = {"workload": desc, "n_rays": n_rays, "kernel": meta, "counters_mean_per_launch": {k: sum(v) / len(v) for k, v in sorted(agg.items())}}
    durs = []
    for f in glob.glob(f"{O}/{w}/trace/**/*_kernel_trace.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "k_trace_phased" in r["Kernel_Name"] and mode_tag in r["Kernel_Name"]]
        durs = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in rows][-3:]
    if durs:
        entry["kernel_stats"] = {"dispatches": len(durs), "average_ns": sum(durs) / len(durs), "min_ns": min(durs), "max_ns": max(durs), "source": "un-profiled rocprofv3 --kernel-trace pass of the same command"}
    c = entry["counters_mean_per_launch"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        entry["hbm"] = {"read_bytes_x2": c["FETCH_SIZE"] * 2048, "read_bytes_x1": c["FETCH_SIZE"] * 1024, "write_bytes": c["WRITE_SIZE"] * 1024,
                        "note": "FETCH_SIZE in KiB; x2 per the gfx950 note of MI355X_MICROARCH.md for coalesced streams, x1 for random 64-byte gathers (profiles/r02_fetch_calibration.txt): the truth lies between"}
    out["workloads"][w] = entry
json.dump(out, open(f"{P}/{R}_pmc_workloads_kernel5.json", "w"), indent=1)
print(json.dumps({w: {"n": len(e["counters_mean_per_launch"]), "avg_ns": e.get("kernel_stats", {}).get("average_ns")} for w, e in out["workloads"].items()}, indent=1))
PY
