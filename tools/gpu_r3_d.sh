#!/bin/bash
# round 3, call D: hardened GPU suite, CPU scaling probe, sub-phase pass counters, default bench line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d; mkdir -p $O
timeout 900 python3 tools/cpu_scaling_probe.py > $O/cpu_scaling.txt 2>&1; cat $O/cpu_scaling.txt
timeout 600 python3 tools/phase_passes.py > $O/r03_phase_passes_kernel5.json 2> $O/phase_passes.err; tail -3 $O/phase_passes.err
timeout 1800 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -8 $O/pytest.log
timeout 900 python3 bench.py --no-extras > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
