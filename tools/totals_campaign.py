"""Campaign: rc_view_factor_totals against the ORACLE's matrix on a scene too large for the default suite (20 k triangles x 1024 rays = 20 M
rays; the oracle needs ~10 s), incl. 2- and 3-way ray partitions and the matrix entry point's own sums."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import raycore_jl_amd as rc
from oracle import pyoracle as po
from helpers import build_oracle, build_product
cfg = rc.scenes.config_c5(lon=60, bands=33, wall_k=8)
t, o = build_product(rc, cfg), build_oracle(po, cfg)
n, rpt = t.n_primitives(), 1024
t0 = time.time(); want = o.view_factors(rpt, seed=21, nthreads=16); dt = time.time() - t0
recv, emit = rc.view_factor_totals(t, rpt, seed=21)
ok = np.array_equal(recv, want.sum(axis=0, dtype=np.uint64)) and np.array_equal(emit, want.sum(axis=1, dtype=np.uint64))
others = [build_product(rc, cfg) for _ in range(2)]
r3, e3 = rc.view_factor_totals_multi([t] + others, rpt, seed=21)
m = rc.view_factors(t, rpt, seed=21)
print(f"{n} triangles x {rpt} rays: totals {t.last_kernel_ms():.1f} ms, oracle matrix {dt:.1f} s; counted {int(recv.sum())}; totals == oracle column / row sums: {ok}; "
      f"3-way ray partition identical: {np.array_equal(r3, recv) and np.array_equal(e3, emit)}; product matrix == oracle matrix: {np.array_equal(m, want)}")
assert ok
