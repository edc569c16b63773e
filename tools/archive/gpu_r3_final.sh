#!/bin/bash
# round 3, final evidence: full GPU suite, phase pass counters, rocprofv3 stats + PMC passes of the bench command + the bench line, HBM regime
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3final; mkdir -p $O
timeout 1200 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
timeout 600 python3 tools/phase_passes.py > $O/r03_phase_passes_kernel5.json 2> $O/phase_passes.err; tail -2 $O/phase_passes.err
timeout 2400 bash tools/capture_profiles.sh r03 > $O/capture.log 2>&1; tail -c 600 $O/capture.log
timeout 1500 bash tools/gpu_hbm_regime.sh r03 > $O/hbm.log 2>&1; tail -c 900 $O/hbm.log
