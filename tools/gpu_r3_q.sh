#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3q; mkdir -p $O
SECONDS=0; python3 bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err | head -2; echo "bench wall time ${SECONDS}s"; python3 - <<'PY'
import json
b=json.loads(open("gpurun_out/r3q/bench.json").read().strip().splitlines()[-1])
print(b["value"], b["ms_per_step"]); r=b["roofline"]; print({k: r[k] for k in r if k not in ("sources","algorithmic_vs_hbm")})
print(b["cpu_baseline"]["value"], b["cpu_baseline"]["cores"])
print([k for k in b["extras"] if k.endswith("_error")])
PY
