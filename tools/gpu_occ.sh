#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02occ; mkdir -p $O
python3 tools/perf_probe.py --workloads c2,c3,shadow,r1m --variants "kernel=5;kernel=5,blocks_per_cu=1;kernel=5,blocks_per_cu=1,pool=64;kernel=5,blocks_per_cu=1,pool=256;kernel=5,blocks_per_cu=1,refill=8;kernel=5" > $O/occ.log 2>&1; cat $O/occ.log
