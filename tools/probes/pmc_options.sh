#!/bin/bash
# dev: SQ / TA / TD counters of the trace kernel under option sets of the product library, one process per (option set, workload, counter set):
#   tools/probes/pmc_options.sh "stack16=0 stack16=1" "c3 c4"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
VS=$1; WS=$2
O=gpurun_out/pmc_opt; rm -rf $O; mkdir -p $O
for v in $VS; do
  for W in $WS; do
    export RC_PROBE_REPS=7
    CMD="python3 tools/perf_probe.py --variants $v --workloads $W"
    i=0
    for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
               "SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" \
               "TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
      i=$((i+1))
      timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$v/$W/pmc_$i -- $CMD > $O/$v.$W.pmc_$i.log 2>&1
    done
  done
done
python3 - "$O" "$VS" "$WS" <<'PY'
import csv, glob, sys, collections
O, VS, WS = sys.argv[1], sys.argv[2].split(), sys.argv[3].split()
for W in WS:
    table = {}
    for v in VS:
        agg = collections.defaultdict(list)
        for f in glob.glob(f"{O}/{v}/{W}/pmc_*/**/*_counter_collection.csv", recursive=True):
            per = collections.defaultdict(dict)
            for r in csv.DictReader(open(f)):
                if "k_trace_phased" in r["Kernel_Name"]:
                    per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
            for d in sorted(per)[-3:]:
                for k, val in per[d].items():
                    agg[k].append(val)
        table[v] = {k: sum(x) / len(x) for k, x in agg.items()}
    keys = sorted(set().union(*[set(t) for t in table.values()]))
    print(f"== {W}: mean of the last 3 dispatches of the trace kernel")
    print(f"{'counter':30s}" + "".join(f"{v:>16s}" for v in VS))
    for k in keys:
        print(f"{k:30s}" + "".join(f"{table[v].get(k, float('nan')):16.4g}" for v in VS))
PY
