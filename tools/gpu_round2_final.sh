#!/bin/bash
# Round-2 final evidence: GPU suite, smoke, profile capture on the final code, the default bench line with extras.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02final; mkdir -p $O
python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
python3 __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash tools/capture_profiles.sh r02final > $O/capture.log 2>&1; tail -c 1500 $O/capture.log
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 6000 $O/bench_default.json
