"""dev: C3 with and without the single BLAS's top nodes in LDS (option blas_top, takes effect at the structural sync): how much do the global node fetches cost now?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raycore_jl_amd as rc
sc = rc.scenes
cfg = sc.config_c3()
rays = sc.c3_primary_rays(cfg, 2048, 2048)
dr = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
dh = torch.empty(len(rays) * 32, dtype=torch.uint8, device="cuda")
for rep in range(2):
    for top in (1, 0):
        t = rc.TLAS(0)
        t.set_option("blas_top", top)
        for v, m in cfg["blas"]: t.add_geometry(v, m)
        for b, xf, ids in cfg["instances"]: t.push_instances(b, xf, ids)
        t.sync()
        best = 1e9
        for _ in range(6):
            t.trace_device(dr.data_ptr(), dh.data_ptr(), len(rays))
            best = min(best, t.last_kernel_ms())
        print(f"blas_top={top}: blas_top_k={t.get_option('blas_top_k')}  C3 {best:.3f} ms  {len(rays)/best/1e3:.0f} Mrays/s", flush=True)
        t.free()
