#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02d; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o /tmp/valu_probe && /tmp/valu_probe 2>&1 | head -8 > $O/valu_masked.txt; cat $O/valu_masked.txt
python3 tools/perf_probe.py --variants "kernel=-1;kernel=5;kernel=3;kernel=6" --workloads c2,c3,shadow,c4,r1m > $O/perf.txt 2>&1; cat $O/perf.txt
python3 -m pytest tests -x -q -m gpu -k "parity or deep or c5 or claims or pool or fuzz" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python3 bench.py --steps 20 --warmup 5 --no-extras > $O/bench.json 2> $O/bench.err; tail -c 1200 $O/bench.json
