#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3r; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_c5_and_claims.py -x -q -m gpu -k "hipgraph" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -25 $O/pytest.log
