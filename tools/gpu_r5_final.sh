#!/bin/bash
# round 5 evidence: full GPU suite, phase pass counters, rocprofv3 stats + PMC passes of the bench command + the bench line (capture_profiles.sh),
# per-workload counters incl. the HBM-bound regime (pmc_workloads.sh), marker + kernel trace of the bench with extras (range_stats.py).
# Copy gpurun_out/r05/profiles/* and gpurun_out/r5final/r05_* to profiles/, then (no GPU):
#   python3 tools/isa_mix.py > profiles/r05_isa_mix_kernel5.json ; python3 tools/range_stats.py gpurun_out/r5final/ranges > profiles/r05_bench_with_extras_by_range.csv
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5final; mkdir -p $O
timeout 1200 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 600 python3 tools/phase_passes.py > $O/r05_phase_passes_kernel5.json 2> $O/phase_passes.err; tail -2 $O/phase_passes.err
timeout 2400 bash tools/capture_profiles.sh r05 > $O/capture.log 2>&1; tail -c 400 $O/capture.log
timeout 2400 bash tools/pmc_workloads.sh r05 > $O/workloads.log 2>&1; tail -c 700 $O/workloads.log
rm -rf $O/ranges; timeout 900 rocprofv3 --marker-trace --kernel-trace --output-format csv -d $O/ranges -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/ranges_bench.json 2> $O/ranges_bench.err
# the bench record with the fresh counter files in place (the box's copy of profiles/ only: the repo's is updated by hand from gpurun_out/)
cp gpurun_out/r05/profiles/r05_pmc_c3.json gpurun_out/r05/profiles/r05_pmc_workloads_kernel5.json profiles/
timeout 900 python3 bench.py > $O/bench_final.json 2> $O/bench_final.err; tail -c 200 $O/bench_final.err
