"""Dev tool: time the trace kernels on the BASELINE configs (device-resident rays, HIP-event kernel time)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raycore_jl_amd as rc
from raycore_jl_amd._capi import check, lib, ptr


def to_dev(a):
    return torch.from_numpy(a.view(np.uint8).reshape(-1)).cuda()


def time_trace(t, rays, mode, reps=5):
    n = len(rays)
    d_rays = to_dev(rays)
    d_hits = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
    best = 1e9
    for _ in range(reps):
        t.trace_device(d_rays.data_ptr(), d_hits.data_ptr(), n, mode=mode)
        best = min(best, t.last_kernel_ms())
    hits = d_hits.cpu().numpy().view(rc.HIT_DT)
    return best, hits


def build(cfg):
    t = rc.TLAS(0)
    for verts, meta in cfg["blas"]:
        t.add_geometry(verts, meta)
    for b, xf, ids in cfg["instances"]:
        t.push_instances(b, xf, ids)
    t0 = time.time()
    t.sync()
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernels", default="0,1")
    ap.add_argument("--bpc", default="0")
    ap.add_argument("--c3res", type=int, default=2048)
    args = ap.parse_args()
    kernels = [int(k) for k in args.kernels.split(",")]
    bpcs = [int(k) for k in args.bpc.split(",")]
    sc = rc.scenes
    cfg2 = sc.config_c2()
    t0 = time.time(); t2 = build(cfg2); print(f"C2 build+sync {time.time()-t0:.3f}s")
    rays2 = rc.generate_ray_grid(t2, cfg2["viewdir"], cfg2["grid"])
    cfg3 = sc.config_c3()
    t0 = time.time(); t3 = build(cfg3); print(f"C3 build+sync {time.time()-t0:.3f}s")
    rays3 = sc.c3_primary_rays(cfg3, args.c3res, args.c3res)
    ms, hits3 = time_trace(t3, rays3, "closest", 1)
    shadow = sc.c3_shadow_rays(cfg3, rays3, hits3)
    bounce = sc.c4_bounce_rays(cfg3, rays3, hits3, 4 * len(rays3))
    for k in kernels:
        for bpc in bpcs:
            for t in (t2, t3):
                t.set_option("kernel", k); t.set_option("blocks_per_cu", bpc)
            for name, t, rays, mode in (("C2 closest", t2, rays2, "closest"), ("C3 primary", t3, rays3, "closest"),
                                        ("C3 shadow-any", t3, shadow, "any"), ("C4 bounce", t3, bounce, "closest")):
                ms, hits = time_trace(t, rays, mode)
                print(f"kernel={k} bpc={bpc} {name:14s} n={len(rays):9d} {ms:9.3f} ms  {len(rays)/ms/1e3:9.1f} Mrays/s  hit={hits['hit'].mean():.3f}", flush=True)


if __name__ == "__main__":
    main()
