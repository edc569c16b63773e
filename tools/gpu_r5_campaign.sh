#!/bin/bash
# Round-5 parity campaigns on the final kernels (claim order worked out inside the launch, 16-bit lane stacks): beyond the default suite.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05camp; mkdir -p $O
timeout 3600 python3 tools/full_parity_campaign.py > $O/full_parity.log 2>&1; tail -8 $O/full_parity.log
RC_FUZZ_SEEDS=27000 timeout 3000 python3 -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -n 0 > $O/fuzz.log 2>&1; tail -3 $O/fuzz.log
RC_STACK16=0 RC_FUZZ_SEEDS=9000 timeout 3000 python3 -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -n 0 > $O/fuzz_stack32.log 2>&1; tail -3 $O/fuzz_stack32.log
timeout 900 python3 tools/totals_campaign.py > $O/totals.log 2>&1; tail -3 $O/totals.log
RC_BENCH_FORCE_DIST=1 timeout 900 python3 bench.py --steps 5 --warmup 2 > $O/bench_force_dist.json 2> $O/bench_force_dist.err; tail -c 300 $O/bench_force_dist.err
RC_BENCH_FORCE_MULTI=2 timeout 900 python3 bench.py --steps 5 --warmup 2 > $O/bench_force_multi.json 2> $O/bench_force_multi.err; tail -c 300 $O/bench_force_multi.err
