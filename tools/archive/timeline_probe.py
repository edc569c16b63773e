"""Dev tool: per-wave timelines of kernel 5 (option "timeline_ptr"): when a wave's claims ran dry, when it was down to 16 / 4 live
lanes, when it ended, how many iterations it ran before and after -- where a mid-size launch's time goes."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import raycore_jl_amd as rc
from tools.perf_probe import build, to_dev


def pct(a, qs=(1, 10, 50, 90, 99, 100)):
    return " ".join(f"p{q}={np.percentile(a, q):8.1f}" for q in qs)


def probe(name, t, rays, mode="closest", opts=None):
    n = len(rays)
    d_rays = to_dev(rays)
    d_hits = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
    t.set_option("kernel", 5)
    for k, v in (opts or {}).items():
        t.set_option(k, v)
    waves = t.get_option("n_cus") * 24
    buf = torch.zeros(waves * 8, dtype=torch.int64, device="cuda")
    for _ in range(2):
        t.trace_device(d_rays.data_ptr(), d_hits.data_ptr(), n, mode=mode)
    plain = t.last_kernel_ms()
    t.set_option("timeline_ptr", buf.data_ptr())
    for _ in range(2):
        t.trace_device(d_rays.data_ptr(), d_hits.data_ptr(), n, mode=mode)
    ms = t.last_kernel_ms()
    t.set_option("timeline_ptr", 0)
    torch.cuda.synchronize()
    w = buf.cpu().numpy().view(np.uint64).reshape(-1, 8)
    w = w[w[:, 4] != 0]
    t0 = w[:, 0].min()
    us = lambda col: (w[:, col].astype(np.int64) - np.int64(t0)) / 100.0  # 100 MHz ticks
    start, tx, t16, t4, tend = us(0), us(1), us(2), us(3), us(4)
    outer, outer_x = (w[:, 5] & 0xFFFFFFFF).astype(np.int64), (w[:, 5] >> 32).astype(np.int64)
    it, it_x = (w[:, 6] & 0xFFFFFFFF).astype(np.int64), (w[:, 6] >> 32).astype(np.int64)
    live_x = w[:, 7].astype(np.int64)
    has_x = w[:, 1] != 0
    print(f"== {name} {opts or ''}: {n} rays, plain {plain:.3f} ms, with timeline {ms:.3f} ms, {len(w)} waves")
    print(f"   wave start          us: {pct(start)}")
    print(f"   claims dry (tx)     us: {pct(tx[has_x])}")
    print(f"   < 16 live lanes     us: {pct(t16[w[:, 2] != 0])}")
    print(f"   <  4 live lanes     us: {pct(t4[w[:, 3] != 0])}")
    print(f"   wave end            us: {pct(tend)}")
    print(f"   interior iterations   : {pct(it)}   of which after tx: {pct(it_x)}")
    dur_x = (tend - tx)[has_x]
    print(f"   drain duration      us: {pct(dur_x)}")
    step_x = dur_x[it_x[has_x] > 20] / it_x[has_x][it_x[has_x] > 20]
    step_b = (tx - start)[has_x & (it - it_x > 20)] / (it - it_x)[has_x & (it - it_x > 20)]
    print(f"   us per interior iteration before tx: {pct(step_b)}")
    print(f"   us per interior iteration after  tx: {pct(step_x)}")
    print(f"   mean live lanes per outer iteration after tx: {live_x.sum() / max(outer_x.sum(), 1):.1f};  outer iterations after tx: {outer_x.sum()} of {outer.sum()}")
    # machine-level: how many waves are still running at time T
    for T in (50, 100, 150, 200, 250, 300, 350, 400):
        print(f"   t={T:3d} us: {int((tend > T).sum()):5d} waves running, {int(((tx > T) | ~has_x).sum()):5d} still have rays to claim", end=";")
    print()
    t.set_option("kernel", -1)


def main():
    sc = rc.scenes
    cfg2 = sc.config_c2()
    t2 = build(cfg2)
    rays2 = rc.generate_ray_grid(t2, cfg2["viewdir"], cfg2["grid"])
    probe("C2", t2, rays2)
    probe("C2", t2, rays2, opts={"pool": 32})
    t2.set_option("pool", 0)
    cfg3 = sc.config_c3()
    t3 = build(cfg3)
    probe("C3 1Mi", t3, sc.c3_primary_rays(cfg3, 1024, 1024))
    probe("C3 4Mi", t3, sc.c3_primary_rays(cfg3, 2048, 2048))


if __name__ == "__main__":
    main()
