"""Dev tool: which kernel wins on small batches (the auto rule's switch point)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raycore_jl_amd as rc
from perf_probe import build, time_trace


def main():
    sc = rc.scenes
    cfg3 = sc.config_c3()
    t3 = build(cfg3)
    cfg2 = sc.config_c2()
    t2 = build(cfg2)
    for name, t, mk in (("C3", t3, lambda r: sc.c3_primary_rays(cfg3, r, r)), ("C2", t2, lambda r: rc.generate_ray_grid(t2, cfg2["viewdir"], r))):
        for res in (64, 128, 181, 256, 362, 512, 724, 886, 1024):
            rays = mk(res)
            out = []
            for k in (0, 1, 3, 5):
                t.set_option("kernel", k)
                ms, _ = time_trace(t, rays, "closest", 8)
                out.append(f"k{k} {ms * 1e3:7.1f} us")
            t.set_option("kernel", -1)
            ms, _ = time_trace(t, rays, "closest", 8)
            print(f"{name} {res:5d}^2 = {len(rays):8d} rays: " + "  ".join(out) + f"  auto {ms * 1e3:7.1f} us", flush=True)


main()
