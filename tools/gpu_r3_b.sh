#!/bin/bash
# round 3, call B: host-matrix view factors (tests + C5 end-to-end timings)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_view_factors_host.py tests/test_gpu_mesh.py tests/test_gpu_c5_and_claims.py tests/test_gpu_reentrant.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -30 $O/pytest.log
timeout 600 python3 tools/vf_e2e_probe.py > $O/vf_e2e.txt 2>&1; cat $O/vf_e2e.txt
