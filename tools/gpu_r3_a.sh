#!/bin/bash
# round 3, call A: full GPU suite on the re-entrant launch bookkeeping + the host-matrix copy probe
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a; mkdir -p $O
timeout 900 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
timeout 300 tools/vf_host_probe > $O/vf_host_probe.txt 2>&1; cat $O/vf_host_probe.txt
timeout 300 python3 tools/perf_probe.py --workloads c2,c3,shadow,c4,r1m --variants "kernel=-1;kernel=-1" > $O/perf.log 2>&1; cat $O/perf.log
