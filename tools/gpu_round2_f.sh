#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02f; mkdir -p $O
V=""
for thr in 36 48 58 64; do for rf in 20 8 1; do V="$V;kernel=5,sched_thr=$thr,refill=$rf"; done; done
V="$V;kernel=5,pool=64;kernel=5,pool=64,sched_thr=64,refill=1;kernel=5,pool=32,sched_thr=64,refill=1;kernel=1;kernel=5,blocks_per_cu=0"
python3 tools/perf_probe.py --variants "${V#;}" --workloads c2,r1m,shadow,c3 > $O/sweep.txt 2>&1; cat $O/sweep.txt
