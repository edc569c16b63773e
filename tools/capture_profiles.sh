#!/bin/bash
# Round-end evidence: rocprofv3 kernel stats + HBM PMC passes of the bench command, summarised under profiles/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${1:-r01}
mkdir -p gpurun_out/$R
CMD="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/stats -- $CMD > gpurun_out/$R/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/$R/pmc_fetch -- $CMD > gpurun_out/$R/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/$R/pmc_write -- $CMD > gpurun_out/$R/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/$R/pmc_l2 -- $CMD > gpurun_out/$R/pmc_l2.log 2>&1
rocprofv3 --pmc TA_TA_BUSY_sum TD_TD_BUSY_sum --kernel-trace --output-format csv -d gpurun_out/$R/pmc_ta -- $CMD > gpurun_out/$R/pmc_ta.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d gpurun_out/$R/pmc_sq -- $CMD > gpurun_out/$R/pmc_sq.log 2>&1
python3 bench.py > gpurun_out/$R/bench.json 2> gpurun_out/$R/bench.err
python3 - "$R" <<'PY'
import csv, glob, json, sys, collections
R = sys.argv[1]
out = {"command": "rocprofv3 --pmc <set> --kernel-trace --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras (one pass per set)"}
agg = collections.defaultdict(list)
meta = {}
for f in glob.glob(f"gpurun_out/{R}/pmc_*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_trace" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "SGPR_Count", "Scratch_Size")}
out["kernel"] = meta
out["counters_mean_per_launch"] = {k: sum(v) / len(v) for k, v in sorted(agg.items())}
c = out["counters_mean_per_launch"]
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    out["hbm"] = {"correction": "FETCH_SIZE x2 (gfx950 reports half of a 16 B/lane coalesced stream, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported; KiB",
                  "read_bytes_per_launch": c["FETCH_SIZE"] * 1024 * 2, "write_bytes_per_launch": c["WRITE_SIZE"] * 1024,
                  "c3_closest_bytes_per_launch": c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024,
                  "expected_stream_bytes": {"rays_in": 4194304 * 32, "hits_out": 4194304 * 32}}
if "TCC_HIT_sum" in c:
    out["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
json.dump(out, open(f"gpurun_out/{R}/pmc_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
for f in glob.glob(f"gpurun_out/{R}/stats/**/*_kernel_stats.csv", recursive=True):
    print(open(f).read()[:1500])
PY
cat gpurun_out/$R/bench.json | tail -1
