#!/bin/bash
# Round-3 parity campaigns on the final kernels (guided + cost-ordered claims, leaf records with edges, SLP off): beyond the default suite.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r03camp; mkdir -p $O
RC_FUZZ_SEEDS=9000 timeout 3000 python3 -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -n 0 > $O/fuzz.log 2>&1; tail -3 $O/fuzz.log
timeout 1500 python3 tools/full_parity_campaign.py > $O/full_parity.log 2>&1; tail -8 $O/full_parity.log
timeout 900 python3 tools/vf_campaign.py > $O/vf.log 2>&1; tail -4 $O/vf.log
timeout 1500 python3 tools/big_blas_campaign.py > $O/big_blas.log 2>&1; tail -6 $O/big_blas.log
