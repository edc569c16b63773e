"""One-off campaign: view_factors at full ray count on a 20k-triangle C5-style scene, GPU matrix vs the oracle's, bit for bit."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import raycore_jl_amd as rc
from oracle import pyoracle as po
from helpers import build_oracle, build_product
cfg = rc.scenes.config_c5(lon=60, bands=33, wall_k=8)
n = len(cfg["blas"][0][0])
rpt = 4096
print("triangles", n, "rays", n * rpt, "matrix MB", 4 * n * n / 1e6, flush=True)
t = build_product(rc, cfg)
t0 = time.time(); got = rc.view_factors(t, rays_per_triangle=rpt, seed=11); print("gpu s", round(time.time() - t0, 3), flush=True)
o = build_oracle(po, cfg)
t0 = time.time(); want = o.view_factors(rpt, seed=11, nthreads=os.cpu_count()); print("oracle s", round(time.time() - t0, 2), flush=True)
print("counted", int(got.sum()), int(want.sum()), "identical:", bool(np.array_equal(got, want)))
