"""Dev tool: does spatial locality of the ray batch matter for a tree larger than one XCD's L2 (C2: 12.8 MB)?  Same 16 M rays traced
as one launch in grid order, as 8 band launches, and in a random order."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raycore_jl_amd as rc
from perf_probe import build, time_trace
sc = rc.scenes
cfg = sc.config_c2()
t = build(cfg)
g = 4096
rays = rc.generate_ray_grid(t, cfg["viewdir"], g)
n = len(rays)
t.set_option("kernel", 3)
ms_all, _ = time_trace(t, rays, "closest", 3)
print(f"one launch, grid order : {ms_all:.3f} ms  {n/ms_all/1e3:.0f} Mrays/s")
tot = 0.0
for b in range(8):
    part = rays[b * n // 8:(b + 1) * n // 8]
    ms, _ = time_trace(t, part, "closest", 3)
    tot += ms
print(f"8 band launches (sum)  : {tot:.3f} ms  {n/tot/1e3:.0f} Mrays/s")
perm = np.random.default_rng(0).permutation(n)
ms_r, _ = time_trace(t, rays[perm], "closest", 3)
print(f"one launch, random order: {ms_r:.3f} ms  {n/ms_r/1e3:.0f} Mrays/s")
# tile order: 64 x 2 pixel tiles (128 rays = one claim) instead of 128 consecutive rays of one column
idx = np.arange(n).reshape(g, g)          # [j][i], ray index = i + g*j
tiles = idx.reshape(g // 2, 2, g // 64, 64).transpose(0, 2, 1, 3).reshape(-1)
ms_t, _ = time_trace(t, rays[tiles], "closest", 3)
print(f"one launch, 64x2 tiles : {ms_t:.3f} ms  {n/ms_t/1e3:.0f} Mrays/s")
tiles = idx.reshape(g // 8, 8, g // 16, 16).transpose(0, 2, 1, 3).reshape(-1)
ms_t, _ = time_trace(t, rays[tiles], "closest", 3)
print(f"one launch, 16x8 tiles : {ms_t:.3f} ms  {n/ms_t/1e3:.0f} Mrays/s")
