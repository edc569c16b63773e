#!/bin/bash
# Round-2 GPU batch A: new tests, microbenchmarks, phase statistics, ray-order probe.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02a; mkdir -p $O
python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o /tmp/valu_probe && /tmp/valu_probe > $O/valu_probe.txt 2>&1; cat $O/valu_probe.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/td_probe.hip -o /tmp/td_probe && /tmp/td_probe > $O/td_probe.txt 2>&1; tail -30 $O/td_probe.txt
python3 tools/perf_probe.py --variants "kernel=-1;kernel=5;kernel=3" --stats --workloads c2,c3,shadow,c4,r1m > $O/perf.txt 2>&1; cat $O/perf.txt
python3 tools/tile_probe.py > $O/tile.txt 2>&1; cat $O/tile.txt
python3 bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
