"""Dev tool: throughput of mid-size batches when the caller keeps several in flight (one stream each): the tail of one launch --
waves waiting for their longest rays -- overlaps the bulk of the next as workgroups of the first exit."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raycore_jl_amd as rc
from tools.perf_probe import build, to_dev


def run(t, rays, n_streams, n_batches, mode="closest"):
    n = len(rays)
    d_rays = to_dev(rays)
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    outs = [torch.empty(n * 32, dtype=torch.uint8, device="cuda") for _ in range(n_streams)]
    for w in range(2):  # warm-up, then timed
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in range(n_batches):
            s = streams[b % n_streams]
            t.trace_device(d_rays.data_ptr(), outs[b % n_streams].data_ptr(), n, mode=mode, stream=s.cuda_stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    ref = outs[0].cpu().numpy().tobytes()
    same = all(o.cpu().numpy().tobytes() == ref for o in outs)
    return dt, same


def main():
    sc = rc.scenes
    cfg2 = sc.config_c2()
    t2 = build(cfg2)
    rays2 = rc.generate_ray_grid(t2, cfg2["viewdir"], cfg2["grid"])
    tb = rc.TLAS(0)
    tb.add_geometry(sc.random_triangles(1_000_000, 42, edge=0.01)); tb.push_instances(1); tb.sync()
    raysb = rc.generate_ray_grid(tb, (0.3, 0.2, 1.0), 1000)
    cfg3 = sc.config_c3()
    t3 = build(cfg3)
    rays3 = sc.c3_primary_rays(cfg3, 1024, 1024)
    for name, t, rays in (("C2 1 M rays", t2, rays2), ("random 1 M tris, 1 M rays", tb, raysb), ("C3 1 Mi primary rays", t3, rays3)):
        for ns in (1, 2, 3, 4):
            nb = 48
            dt, same = run(t, rays, ns, nb)
            print(f"{name:28s} {ns} stream(s) x {nb} batches: {dt / nb * 1e3:.3f} ms per batch  {len(rays) * nb / dt / 1e9:.2f} Grays/s  identical={same}  drift={t.get_option('claim_drift')}")


if __name__ == "__main__":
    main()
