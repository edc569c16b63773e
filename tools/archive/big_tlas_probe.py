"""Dev tool: a top level too large for the LDS kernels (5000 instances of one 4096-triangle BLAS): kernel 3 vs the others."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raycore_jl_amd as rc
from perf_probe import time_trace

sc = rc.scenes
for lattice in ((8, 8, 4), (10, 10, 5), (20, 20, 12)):
    cfg = sc.config_c3(lattice=lattice)
    t = rc.TLAS(0)
    for verts, meta in cfg["blas"]:
        t.add_geometry(verts, meta)
    for b, xf, ids in cfg["instances"]:
        t.push_instances(b, xf, ids)
    t.sync()
    rays = sc.c3_primary_rays(cfg, 2048, 2048)
    out = []
    for k in (-1, 3, 5, 1):
        t.set_option("kernel", k)
        ms, hits = time_trace(t, rays, "closest")
        out.append(f"k{k} {ms:.3f} ms {len(rays) / ms / 1e3:.0f} Mrays/s")
    print(f"{np.prod(lattice)} instances, hit {hits['hit'].mean():.3f}: " + "  ".join(out), flush=True)
    t.free()
