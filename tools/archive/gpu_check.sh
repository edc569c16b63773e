#!/bin/bash
# GPU suite + smoke + the perf probe on the BASELINE workloads (dev loop: run after a kernel change)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-check}; mkdir -p $O
python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
python3 __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python3 tools/perf_probe.py --workloads c2,c3,shadow,c4,r1m --variants "kernel=-1;kernel=-1" > $O/perf.log 2>&1; cat $O/perf.log
