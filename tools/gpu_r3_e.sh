#!/bin/bash
# round 3, call E: tapered (guided) chunk sizes -- sweep on the BASELINE workloads
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e; mkdir -p $O
timeout 1500 python3 tools/perf_probe.py --workloads c2,r1m,shadow,c3,c4 --variants "taper=0;taper=2;taper=4;taper=8;taper=12;taper=16;taper=24;taper=32;taper=8,pool=256;taper=16,pool=256;taper=8,pool=64;taper=0" > $O/taper_sweep.txt 2>&1; cat $O/taper_sweep.txt
