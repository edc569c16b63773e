#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3o; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "edge or weird or nan or c3_c4 or c2_prop or build_parity" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
RC_PROBE_REPS=10 timeout 900 python3 tools/perf_probe.py --workloads c3,c4,shadow,c2 --variants "kernel=-1;kernel=-1;kernel=3" > $O/perf.txt 2>&1; grep "^\[" $O/perf.txt
