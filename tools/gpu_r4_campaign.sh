#!/bin/bash
# Round-4 parity campaigns on the final kernels (batch slots, sparse recording, private accumulators): beyond the default suite.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04camp; mkdir -p $O
RC_FUZZ_SEEDS=9000 timeout 3000 python3 -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -n 0 > $O/fuzz.log 2>&1; tail -3 $O/fuzz.log
timeout 2400 python3 tools/full_parity_campaign.py > $O/full_parity.log 2>&1; tail -8 $O/full_parity.log
timeout 900 python3 tools/totals_campaign.py > $O/totals.log 2>&1; tail -3 $O/totals.log
