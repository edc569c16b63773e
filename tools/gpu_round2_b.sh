#!/bin/bash
# Round-2 GPU batch B: full GPU suite on the self-resetting claim counters, extended VALU probe, perf check, SQ counters of the bench kernel.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02b; mkdir -p $O
python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o /tmp/valu_probe && /tmp/valu_probe > $O/valu_probe.txt 2>&1; cat $O/valu_probe.txt
python3 tools/perf_probe.py --variants "kernel=-1;kernel=5" --workloads c2,c3,shadow,c4,r1m > $O/perf.txt 2>&1; cat $O/perf.txt
CMD="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $CMD > $O/stats.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/pmc_sq -- $CMD > $O/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $O/pmc_sq2 -- $CMD > $O/pmc_sq2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $O/pmc_grbm -- $CMD > $O/pmc_grbm.log 2>&1
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("gpurun_out/r02b/pmc_*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_trace" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k][-10:]
    print(f"{k:32s} n={len(agg[k])} mean_last10={sum(v)/len(v):.5g}")
for f in glob.glob("gpurun_out/r02b/stats/**/*_kernel_stats.csv", recursive=True):
    print(open(f).read()[:1200])
PY
python3 bench.py --steps 20 --warmup 5 --no-extras > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
