#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c; mkdir -p $O
python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o /tmp/valu_probe && /tmp/valu_probe > $O/valu_probe.txt 2>&1; cat $O/valu_probe.txt
python3 __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
bash tools/capture_profiles.sh r02c > $O/capture.log 2>&1; tail -c 6000 $O/capture.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err; tail -c 4000 $O/bench_full.json
