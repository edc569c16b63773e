#!/bin/bash
# One-off parity campaigns on the round-2 kernels (self-resetting claim counters, pointer-form lane stack): beyond the default suite.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02camp; mkdir -p $O
RC_FUZZ_SEEDS=1500 timeout 1500 python3 -m pytest tests/test_gpu_fuzz.py -x -q -m gpu > $O/fuzz.log 2>&1; tail -3 $O/fuzz.log
timeout 1500 python3 tools/full_parity_campaign.py > $O/full_parity.log 2>&1; tail -8 $O/full_parity.log
timeout 900 python3 tools/vf_campaign.py > $O/vf.log 2>&1; tail -4 $O/vf.log
