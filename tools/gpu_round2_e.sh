#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02e; mkdir -p $O
for W in c2 r1m c4 shadow; do
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
             "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
             "TA_TA_BUSY_sum TD_TD_BUSY_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$W/set$i -- python3 tools/perf_probe.py --variants "kernel=5" --workloads $W > $O/$W.set$i.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections, json
res = {}
for W in ("c2", "r1m", "c4", "shadow"):
    agg = collections.defaultdict(list)
    dur = []
    for f in glob.glob(f"gpurun_out/r02e/{W}/set*/**/*_counter_collection.csv", recursive=True):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_trace" in r["Kernel_Name"]:
                per[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in per.items():
            agg[k] += v[-5:]          # the last 5 dispatches = the timed repetitions of the workload
    for f in glob.glob(f"gpurun_out/r02e/{W}/set1/**/*_kernel_trace.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "k_trace" in r["Kernel_Name"]]
        dur = [(float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) for r in rows[-5:]]
    c = {k: sum(v) / len(v) for k, v in agg.items()}
    t = (sum(dur) / len(dur) * 1e-9) if dur else None
    d = {"counters_mean_per_launch": c, "launch_seconds_in_pmc_pass": t}
    if t and "SQ_INSTS_VALU" in c:
        clk = c.get("GRBM_GUI_ACTIVE", 0) / 8 / t if c.get("GRBM_GUI_ACTIVE") else 2.3e9
        d["derived"] = {"clock_hz": clk, "valu_issue_fraction_4cyc": c["SQ_INSTS_VALU"] * 4 / (1024 * clk * t),
                        "lane_utilisation": c.get("SQ_THREAD_CYCLES_VALU", 0) / (c["SQ_INSTS_VALU"] * 64),
                        "td_busy": c.get("TD_TD_BUSY_sum", 0) / (256 * clk * t), "ta_busy": c.get("TA_TA_BUSY_sum", 0) / (256 * clk * t),
                        "l2_hit": c.get("TCC_HIT_sum", 0) / max(c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0), 1),
                        "hbm_bytes": c.get("FETCH_SIZE", 0) * 2048 + c.get("WRITE_SIZE", 0) * 1024}
    res[W] = d
json.dump(res, open("gpurun_out/r02e/pmc_workloads.json", "w"), indent=1)
for W, d in res.items():
    print(W, json.dumps(d.get("derived"), indent=None), "t=", d["launch_seconds_in_pmc_pass"])
PY
