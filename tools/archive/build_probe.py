"""Dev tool: BLAS / TLAS build times (reference publishes 4.93 / 7.46 / 16.16 ms for 250k / 1M / 4M triangles on an
RX 7900 XTX, benchmarks/implicitbvh_comparison.md:12-14)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raycore_jl_amd as rc

for n in (250_000, 1_000_000, 4_000_000):
    verts = rc.scenes.random_triangles(n, 42, edge=0.01)
    t = rc.TLAS(0)
    times = []
    dev = []
    d_verts = torch.from_numpy(verts).cuda()
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b = t.add_geometry(verts)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    for rep in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        b = t.add_geometry_device(d_verts.data_ptr(), n)
        torch.cuda.synchronize()
        dev.append(((time.perf_counter() - t0) * 1e3, t.last_kernel_ms()))
    t.push_instances(1)
    t0 = time.perf_counter()
    t.sync()
    sync_ms = (time.perf_counter() - t0) * 1e3
    rays = rc.generate_ray_grid(t, [0.3, 0.2, 1.0], 1000)
    d_rays = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
    d_hits = torch.empty(len(rays) * 32, dtype=torch.uint8, device="cuda")
    best = 1e9
    for _ in range(5):
        t.trace_device(d_rays.data_ptr(), d_hits.data_ptr(), len(rays))
        best = min(best, t.last_kernel_ms())
    print(f"n={n:8d} add_blas wall ms (host filter + H2D + device LBVH): {[round(x, 2) for x in times]}  device-resident wall/kernels ms {[(round(a, 2), round(b, 2)) for a, b in dev]}  sync {sync_ms:.2f} ms"
          f"  | 1M grid rays closest_hit {best:.3f} ms = {len(rays)/best/1e3:.0f} Mrays/s", flush=True)
    t.free()
