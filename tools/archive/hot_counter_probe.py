"""Dev tool: get_illumination on a scene where most rays land on two large triangles (a ground plane) -- one accumulator word takes
most of the counts."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raycore_jl_amd as rc

sc = rc.scenes
ground = np.array([[-50, -50, 0, 50, -50, 0, 50, 50, 0], [-50, -50, 0, 50, 50, 0, -50, 50, 0]], dtype=np.float32)
sphere = sc.fan_sphere(64, 33, centre=(0, 0, 3), radius=2.0)
verts = np.concatenate([ground, sphere]).astype(np.float32)
t = rc.TLAS(0)
t.push(verts, meta=np.arange(1, len(verts) + 1, dtype=np.uint32))
t.sync()
for grid in (1000, 2000):
    best = 1e9
    for _ in range(4):
        c = rc.get_illumination(t, [0.05, 0.02, -1.0], grid)
        best = min(best, t.last_kernel_ms())
    print(f"grid {grid}: kernel {best:.3f} ms, {grid * grid / best / 1e3:.0f} Mrays/s; counts on the two ground triangles {c[0]:.0f} + {c[1]:.0f} of {c.sum():.0f}")
