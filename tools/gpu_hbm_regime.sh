#!/bin/bash
# HBM-bound regime: physical HBM bytes (rocprofv3 FETCH_SIZE / WRITE_SIZE, separate passes) vs algorithmic bytes.
#   tools/gpu_hbm_regime.sh r03      -> gpurun_out/r03hbm/r03_hbm_regime.json (copy to profiles/; bench.py reads exactly these keys)
# Calibration (profiles/r02_fetch_calibration.txt): FETCH_SIZE counts a random 64-byte record once (x1) and a coalesced 16 B/lane
# stream at half (x2, the gfx950 note of MI355X_MICROARCH.md).  In this regime the reads are node gathers (x1) plus the ray stream
# (32 B per ray, coalesced: reported at half, so rays x 16 bytes are added back).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${1:-r03}
O=gpurun_out/${R}hbm; mkdir -p $O
python3 tools/hbm_regime.py --out $O/regime.json > $O/regime.log 2>&1; cat $O/regime.log | tail -4
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-20)
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$tag -- python3 tools/hbm_regime.py --no-stats > $O/pmc_$tag.log 2>&1
done
python3 - "$R" <<'PY'
import csv, glob, json, collections, sys
R = sys.argv[1]
O = f"gpurun_out/{R}hbm"
reg = json.load(open(f"{O}/regime.json"))
# per scene: the 5 timed launches of the default kernel are consecutive k_trace dispatches; scenes come in the order of regime.json
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{O}/pmc_*/**/*_counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "k_trace" in r["Kernel_Name"]]
    by_counter = collections.defaultdict(list)
    for r in rows:
        by_counter[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, vals in by_counter.items():
        k = len(vals) // len(reg)
        for i, key in enumerate(reg):
            per[key][name] = vals[i * k:(i + 1) * k]
out = {"what": "The trace kernel (default kernel, closest_hit) in an HBM-bound regime: one BLAS of random small triangles far larger than L2 + Infinity Cache, 4 194 304 incoherent rays (tools/hbm_regime.py under tools/gpu_hbm_regime.sh)",
       "method": "rocprofv3 --pmc <counter> --kernel-trace, one pass per counter set, mean of the 5 timed launches; kernel time from HIP events of an un-profiled run; FETCH_SIZE x1 for the node gathers (random 64-byte records, profiles/r02_fetch_calibration.txt) + rays x 16 B for the half-reported coalesced ray stream; WRITE_SIZE as reported (KiB)",
       "hbm_peak_GBs": 8000.0, "scenes": {}}
for key, e in reg.items():
    c = {k: sum(v) / len(v) for k, v in per[key].items() if v}
    rd = c.get("FETCH_SIZE", 0) * 1024 + e["rays"] * 16
    wr = c.get("WRITE_SIZE", 0) * 1024
    t = e["ms"] * 1e-3
    e2 = {"triangles": e["triangles"], "node_array_bytes": e["node_bytes"], "rays": e["rays"], "hit_fraction": e["hit_fraction"], "launch_ms": e["ms"], "mrays_s": e["mrays_s"],
          "node_fetches_per_ray": e.get("node_fetches_per_ray"), "algorithmic_bytes_per_launch": e.get("algorithmic_bytes_per_launch"),
          "algorithmic_GBs": e.get("algorithmic_bytes_per_launch", 0) / t / 1e9, "counters_mean_per_launch": c,
          "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_physical_GBs": (rd + wr) / t / 1e9, "hbm_physical_frac_of_8TBs": (rd + wr) / t / 1e9 / 8000.0,
          "memory_requests_per_second_G": c.get("TCC_EA0_RDREQ_sum", 0) / t / 1e9 if c.get("TCC_EA0_RDREQ_sum") else None,
          "fetch_amplification": (rd + wr) / e["algorithmic_bytes_per_launch"] if e.get("algorithmic_bytes_per_launch") else None,
          "l2_hit_rate": c.get("TCC_HIT_sum", 0) / max(c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0), 1),
          "valu_wave_instructions_per_launch": c.get("SQ_INSTS_VALU"),
          "wave_cycles_waiting_fraction": c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else None}
    out["scenes"][key] = e2
json.dump(out, open(f"{O}/{R}_hbm_regime.json", "w"), indent=1)
print(json.dumps({k: {kk: v[kk] for kk in ("mrays_s", "hbm_physical_GBs", "hbm_physical_frac_of_8TBs", "algorithmic_GBs", "fetch_amplification", "l2_hit_rate")} for k, v in out["scenes"].items()}, indent=1))
PY
