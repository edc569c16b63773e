"""Dev tool: where view_factors' time goes -- the same rays traced from a buffer (no generation, no matrix atomics) vs the fused driver."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raycore_jl_amd as rc
from raycore_jl_amd._capi import check, lib, ptr

sc = rc.scenes
cfg = sc.config_c5()
t = rc.TLAS(0)
t.add_geometry(*cfg["blas"][0])
t.push_instances(1, cfg["instances"][0][1], cfg["instances"][0][2])
t.sync()
n = t.n_primitives()
rpt, n_src = 4096, 2048
first = n // 2 - n_src // 2  # a block of sources in the middle of the sorted array
rays = torch.empty(n_src * rpt * 32, dtype=torch.uint8, device="cuda")
for k in range(n_src):
    check(lib().rc_view_factor_rays_device(t._h, 7, first + k, 0, rpt, ptr(rays.data_ptr() + k * rpt * 32), None))
torch.cuda.synchronize()
hits = torch.empty(n_src * rpt * 32, dtype=torch.uint8, device="cuda")
best = 1e9
for _ in range(5):
    t.trace_device(rays.data_ptr(), hits.data_ptr(), n_src * rpt)
    best = min(best, t.last_kernel_ms())
print(f"trace of {n_src * rpt} generated rays from a buffer: {best:.3f} ms = {n_src * rpt / best / 1e3:.0f} Mrays/s; hit fraction {hits.cpu().numpy().view(rc.HIT_DT)['hit'].mean():.3f}")
m = torch.zeros(n_src * n, dtype=torch.int32, device="cuda")
best = 1e9
for _ in range(5):
    check(lib().rc_view_factors_device(t._h, rpt, 7, first, first + n_src, 0, rpt, ptr(m.data_ptr()), n, 1, first, 1, None))
    torch.cuda.synchronize()
    best = min(best, t.last_kernel_ms())
print(f"fused view_factors on the same sources: {best:.3f} ms = {n_src * rpt / best / 1e3:.0f} Mrays/s")
