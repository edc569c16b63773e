#!/bin/bash
# Round-6 parity campaigns on the final kernels (ray-range touch at the claim, pause word with its launch number, batches entry point): beyond the default suite.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06camp; mkdir -p $O
timeout 3600 python3 tools/full_parity_campaign.py > $O/full_parity.log 2>&1; tail -8 $O/full_parity.log
RC_FUZZ_SEEDS=27000 timeout 3000 python3 -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -n 0 > $O/fuzz.log 2>&1; tail -3 $O/fuzz.log
RC_STACK16=0 RC_FUZZ_SEEDS=9000 timeout 3000 python3 -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -n 0 > $O/fuzz_stack32.log 2>&1; tail -3 $O/fuzz_stack32.log
timeout 900 python3 tools/totals_campaign.py > $O/totals.log 2>&1; tail -3 $O/totals.log
RC_BENCH_FORCE_DIST=1 timeout 900 python3 bench.py --steps 5 --warmup 2 > $O/bench_force_dist.json 2> $O/bench_force_dist.err; tail -c 300 $O/bench_force_dist.err
RC_BENCH_FORCE_MULTI=2 timeout 900 python3 bench.py --steps 5 --warmup 2 > $O/bench_force_multi.json 2> $O/bench_force_multi.err; tail -c 300 $O/bench_force_multi.err
# round 6: the multi-rank branches against the stub communicator at C5 size (one child: 8 ranks on device 0, totals only -- the matrix would be 8 x 10 GB)
python3 - > $O/fake_rccl_c5_totals.log 2>&1 <<'PY'
import ctypes as C, os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from fake_rccl.build import build
so = build()
os.environ["RC_RCCL_LIBRARY"] = so; os.environ["RC_ENABLE_DEBUG_HOOKS"] = "1"; os.environ["RC_DEBUG_RANKS_SHARE_DEVICE"] = "1"
import numpy as np
import raycore_jl_amd as rc
from helpers import build_product
cfg = rc.scenes.config_c5()
scenes = [build_product(rc, cfg) for _ in range(8)]
rpt = cfg["rays_per_triangle"]
prep = rc.multi_prepare(scenes)
r1, e1 = rc.view_factor_totals(scenes[0], rpt, seed=7)
t0 = time.time(); r8, e8 = rc.view_factor_totals_multi(scenes, rpt, seed=7); dt = time.time() - t0
fake = C.CDLL(so); out = (C.c_uint64 * 6)(); fake.fake_rccl_stats.argtypes = [C.POINTER(C.c_uint64)]; fake.fake_rccl_stats(out)
print(f"C5 ({scenes[0].n_primitives()} triangles x {rpt} rays) totals on 8 ranks sharing device 0 through the stub communicator: rccl_ranks {prep['rccl_ranks']}, "
      f"identical to one scene's totals: {bool(np.array_equal(r1, r8) and np.array_equal(e1, e8))}, rays counted {int(r8.sum())}, wall {dt * 1e3:.1f} ms; stub saw {list(out)} (worlds, calls, collectives, elements, group launches, max ranks)")
PY
tail -2 $O/fake_rccl_c5_totals.log
