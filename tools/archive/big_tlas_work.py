"""Is the lower rate of the >256-instance scenes (trace kernel 6) a worse kernel or more work per ray?  Per-ray visit counts (kernel 3's
STATS build: the reference algorithm's visits, identical for every kernel) and rates of kernels 3 / 5 / 6 on the C3 BLAS at 256, 500 and
4 800 instances, same 2048 x 2048 pinhole camera."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raycore_jl_amd as rc
from tools.perf_probe import build, to_dev
sc = rc.scenes
for lattice in ((8, 8, 4), (10, 10, 5), (20, 20, 12)):
    cfg = sc.config_c3(lattice=lattice)
    t = build(cfg)
    rays = sc.c3_primary_rays(cfg, 2048, 2048)
    n = len(rays)
    d_r, d_h = to_dev(rays), torch.empty(n * 32, dtype=torch.uint8, device="cuda")
    t.set_option("kernel", 3); t.set_option("stats", 1)
    t.trace_device(d_r.data_ptr(), d_h.data_ptr(), n); torch.cuda.synchronize()
    v = [t.get_option(f"stat{i}") for i in range(8)]
    t.set_option("stats", 0)
    line = f"{int(np.prod(lattice)):5d} instances: per ray {v[3] / n:6.2f} interior visits, {v[5] / n:5.2f} leaf tests, {v[7] / n:5.2f} level switches (entries + exits); fill I {v[3] / max(v[2], 1):4.1f} L {v[5] / max(v[4], 1):4.1f} S {v[7] / max(v[6], 1):4.1f}"
    rates = {}
    for k in (3, 5, 6, -1):
        t.set_option("kernel", k)
        best = 1e9
        for _ in range(6):
            t.trace_device(d_r.data_ptr(), d_h.data_ptr(), n)
            best = min(best, t.last_kernel_ms())
        rates[k] = n / best / 1e3
    hit = d_h.cpu().numpy().view(rc.HIT_DT)["hit"].mean()
    work = v[3] / n + 1.7 * v[5] / n + 1.3 * v[7] / n  # interior-visit equivalents (a leaf pass costs ~82 / 48, an entry ~65 / 48 of an interior step)
    print(line + f"; hit {hit:.3f}; Mrays/s kernel 3 / 5 / 6 / auto: {rates[3]:.0f} / {rates[5]:.0f} / {rates[6]:.0f} / {rates[-1]:.0f}; "
          f"work {work:.1f} interior-visit equivalents per ray => auto kernel {rates[-1] * work / 1e3:.1f} G visit-equivalents / s   tlas_top_k {t.get_option('tlas_top_k')} blas_top_k {t.get_option('blas_top_k')}", flush=True)
    t.free()
