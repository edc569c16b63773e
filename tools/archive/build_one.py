"""Dev tool (profiling target): N device-resident BLAS builds of one size.  usage: build_one.py [n_triangles] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raycore_jl_amd as rc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
verts = rc.scenes.random_triangles(n, 42, edge=0.01)
d_verts = torch.from_numpy(verts).cuda()
t = rc.TLAS(0)
ms = []
for _ in range(reps):
    t.add_geometry_device(d_verts.data_ptr(), n)
    torch.cuda.synchronize()
    ms.append(t.last_kernel_ms())
print(f"n={n} device build ms: min {min(ms):.3f} median {sorted(ms)[len(ms) // 2]:.3f}")
