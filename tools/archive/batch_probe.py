"""Dev tool: fixed cost vs per-ray cost of one trace launch -- the same ray set repeated k times in one batch (same coherence)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raycore_jl_amd as rc
from perf_probe import build, to_dev


def main():
    sc = rc.scenes
    cfg2 = sc.config_c2()
    t2 = build(cfg2)
    rays2 = rc.generate_ray_grid(t2, cfg2["viewdir"], cfg2["grid"])
    cfg3 = sc.config_c3()
    t3 = build(cfg3)
    rays3 = sc.c3_primary_rays(cfg3, 1024, 1024)
    for name, t, rays in (("C2 1M grid", t2, rays2), ("C3 1M primary", t3, rays3)):
        for kern in (-1, 0):
            t.set_option("kernel", kern)
            for k in (1, 2, 4, 8):
                for order in ("concat", "interleave"):
                    if k == 1 and order == "interleave":
                        continue
                    r = np.concatenate([rays] * k) if order == "concat" else np.repeat(rays, k)
                    d = to_dev(r)
                    out = torch.empty(len(r) * 32, dtype=torch.uint8, device="cuda")
                    best = 1e9
                    for _ in range(6):
                        t.trace_device(d.data_ptr(), out.data_ptr(), len(r), mode="closest")
                        best = min(best, t.last_kernel_ms())
                    print(f"{name} kernel={kern} x{k} {order:10s} n={len(r):9d} {best:.3f} ms  {len(r) / best / 1e3:.0f} Mrays/s", flush=True)


main()
