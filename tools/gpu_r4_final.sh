#!/bin/bash
# round 4 evidence: full GPU suite, phase pass counters, rocprofv3 stats + PMC passes of the bench command + the bench line (capture_profiles.sh),
# per-workload counters (pmc_workloads.sh).  Copy gpurun_out/r04/profiles/* and gpurun_out/r4final/r04_phase_passes_kernel5.json to profiles/,
# then (no GPU): python3 tools/isa_mix.py --passes profiles/r04_phase_passes_kernel5.json --counters profiles/r04_pmc_c3.json > profiles/r04_isa_mix_kernel5.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4final; mkdir -p $O
timeout 1200 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -3 $O/pytest.log
timeout 600 python3 tools/phase_passes.py > $O/r04_phase_passes_kernel5.json 2> $O/phase_passes.err; tail -2 $O/phase_passes.err
timeout 2400 bash tools/capture_profiles.sh r04 > $O/capture.log 2>&1; tail -c 400 $O/capture.log
timeout 2400 bash tools/pmc_workloads.sh r04 > $O/workloads.log 2>&1; tail -c 600 $O/workloads.log
