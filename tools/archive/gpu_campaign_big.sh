#!/bin/bash
# The long parity campaigns (round 1's sizes) on the current kernels.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02campbig; mkdir -p $O
RC_FUZZ_SEEDS=9000 timeout 2400 python3 -m pytest tests/test_gpu_fuzz.py -x -q -m gpu > $O/fuzz.log 2>&1; tail -3 $O/fuzz.log
timeout 1500 python3 tools/big_blas_campaign.py > $O/big_blas.log 2>&1; tail -6 $O/big_blas.log
