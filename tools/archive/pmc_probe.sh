#!/bin/bash
# Dev tool: collect SQ / cache counters for the C3 closest_hit launch (separate rocprofv3 --pmc passes).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
rocprofv3 -L > gpurun_out/pmc/counters_list.txt 2>&1
CMD="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVES_EQ_64 SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc/set$i -- $CMD > gpurun_out/pmc/set$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc/set*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_trace" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("gpurun_out/pmc/summary.txt", "w") as o:
    for k in sorted(agg):
        v = agg[k]
        o.write(f"{k:40s} n={len(v)} mean={sum(v)/len(v):.4g}\n")
print(open("gpurun_out/pmc/summary.txt").read())
PY
