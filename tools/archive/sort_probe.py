"""Dev tool: does ordering an incoherent ray batch (C4 bounce rays, C3 shadow rays) by a spatial key pay for itself?"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raycore_jl_amd as rc
from perf_probe import build, time_trace


def expand(v):
    v = v.astype(np.uint64)
    v = (v | (v << 32)) & 0x1F00000000FFFF
    v = (v | (v << 16)) & 0x1F0000FF0000FF
    v = (v | (v << 8)) & 0x100F00F00F00F00F
    v = (v | (v << 4)) & 0x10C30C30C30C30C3
    v = (v | (v << 2)) & 0x1249249249249249
    return v


def keys(rays, bits, with_dir):
    o = rays["o"].astype(np.float64)
    lo, hi = o.min(0), o.max(0)
    q = np.clip((o - lo) / (hi - lo) * (1 << bits), 0, (1 << bits) - 1).astype(np.uint64)
    k = (expand(q[:, 0]) << np.uint64(2)) | (expand(q[:, 1]) << np.uint64(1)) | expand(q[:, 2])
    if with_dir == "octant_hi":
        dd = rays["d"]
        octant = ((dd[:, 0] < 0).astype(np.uint64) << np.uint64(2)) | ((dd[:, 1] < 0).astype(np.uint64) << np.uint64(1)) | (dd[:, 2] < 0).astype(np.uint64)
        k |= octant << np.uint64(60)
    elif with_dir == "octant_lo":
        dd = rays["d"]
        octant = ((dd[:, 0] < 0).astype(np.uint64) << np.uint64(2)) | ((dd[:, 1] < 0).astype(np.uint64) << np.uint64(1)) | (dd[:, 2] < 0).astype(np.uint64)
        k = (k << np.uint64(3)) | octant
    return k


def main():
    sc = rc.scenes
    cfg3 = sc.config_c3()
    t3 = build(cfg3)
    rays3 = sc.c3_primary_rays(cfg3, 2048, 2048)
    _, hits3 = time_trace(t3, rays3, "closest", 1)
    shadow = sc.c3_shadow_rays(cfg3, rays3, hits3)
    bounce = sc.c4_bounce_rays(cfg3, rays3, hits3, 4 * len(rays3))
    for name, rays, mode in (("C4 bounce", bounce, "closest"), ("C3 shadow", shadow, "any")):
        ms, _ = time_trace(t3, rays, mode)
        print(f"{name} as generated: {ms:.3f} ms {len(rays) / ms / 1e3:.0f} Mrays/s", flush=True)
        rng = np.random.default_rng(1)
        ms, _ = time_trace(t3, rays[rng.permutation(len(rays))], mode)
        print(f"{name} shuffled: {ms:.3f} ms {len(rays) / ms / 1e3:.0f} Mrays/s", flush=True)
        for bits in (4, 6, 10):
            for wd in ("none", "octant_hi", "octant_lo"):
                k = keys(rays, bits, wd)
                r = rays[np.argsort(k, kind="stable")]
                ms, _ = time_trace(t3, r, mode)
                print(f"{name} sorted by {bits}-bit origin morton, dir={wd}: {ms:.3f} ms {len(rays) / ms / 1e3:.0f} Mrays/s", flush=True)


main()
