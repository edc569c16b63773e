#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_c5_and_claims.py tests/test_gpu_reentrant.py tests/test_gpu_parity.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log
timeout 1500 python3 tools/perf_probe.py --workloads c2,r1m,shadow,c3,c4 --variants "taper=0,cost_order=0;taper=12,cost_order=0;taper=12,cost_order=1;taper=12,cost_order=1,cost_thr=4;taper=12,cost_order=1,cost_thr=16;taper=8,cost_order=1;taper=16,cost_order=1;taper=12,cost_order=1" > $O/cost_order_sweep.txt 2>&1; cat $O/cost_order_sweep.txt
