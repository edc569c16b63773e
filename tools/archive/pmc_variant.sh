#!/bin/bash
# usage: tools/pmc_variant.sh "<variant>" <workload> <tag>   -- SQ / TA / TCP counters for one kernel variant
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
V="$1"; W="$2"; TAG="$3"
mkdir -p gpurun_out/pmcv/$TAG
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TA_TA_BUSY_sum TA_BUSY_avr" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TA_BUFFER_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum" "TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmcv/$TAG/set$i -- python3 tools/perf_probe.py --variants "$V" --workloads $W > gpurun_out/pmcv/$TAG/set$i.log 2>&1
done
python3 - "$TAG" <<'PY'
import csv, glob, collections, sys
tag = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/pmcv/{tag}/set*/**/*_counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "k_trace" in r["Kernel_Name"]]
    per = collections.defaultdict(list)
    for r in rows:
        per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in per.items():
        agg[k] += v[-5:]   # the last 5 dispatches are the timed reps of the requested variant
with open(f"gpurun_out/pmcv/{tag}/summary.txt", "w") as o:
    for k in sorted(agg):
        v = agg[k]
        o.write(f"{k:40s} n={len(v)} mean={sum(v)/len(v):.4g}\n")
print(tag); print(open(f"gpurun_out/pmcv/{tag}/summary.txt").read())
PY
