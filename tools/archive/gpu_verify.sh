#!/bin/bash
# End-of-round verification on the GPU box: the GPU suite, smoke(), the default bench line.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/verify; mkdir -p $O
python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
python3 __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python3 bench.py --no-extras > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print({k: d[k] for k in ('value','ms_per_step','n_gpus')}, d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['gpu_matches_bit_exact'])"
