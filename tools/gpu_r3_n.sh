#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3n; mkdir -p $O
RC_PROBE_REPS=12 timeout 1500 python3 tools/perf_probe.py --workloads c2,r1m,shadow,c3 --variants "pool=128,taper=12;pool=64,taper=12;pool=64,taper=8;pool=64,taper=16;pool=64,taper=24;pool=256,taper=12;pool=256,taper=16;pool=128,taper=12,refill=12;pool=128,taper=12,refill=28;pool=128,taper=12,sched_thr=30;pool=128,taper=12,sched_thr=42;pool=128,taper=12" > $O/pool_taper_sweep.txt 2>&1; cat $O/pool_taper_sweep.txt
