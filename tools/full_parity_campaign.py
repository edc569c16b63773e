"""One-off campaign: every ray of the BASELINE C2 / C3 / shadow / C4 workloads, every kernel variant, against the oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import raycore_jl_amd as rc
from oracle import pyoracle as po
from helpers import build_oracle, build_product, assert_hits_equal
sc = rc.scenes
nt = 16  # the GPU box grants 16 CPUs of time (cgroup quota), whatever it shows
cfg3 = sc.config_c3()
t3, o3 = build_product(rc, cfg3), build_oracle(po, cfg3)
rays = sc.c3_primary_rays(cfg3, 2048, 2048)
want = o3.trace(rays, nthreads=nt)
shadow = sc.c3_shadow_rays(cfg3, rays, want)
bounce = sc.c4_bounce_rays(cfg3, rays, want, 4 * len(rays))
cfg2 = sc.config_c2()
t2, o2 = build_product(rc, cfg2), build_oracle(po, cfg2)
rays2 = rc.generate_ray_grid(t2, cfg2["viewdir"], cfg2["grid"])
work = [("C3 primary closest", t3, o3, rays, "closest"), ("C3 shadow any", t3, o3, shadow, "any"), ("C4 bounce closest", t3, o3, bounce, "closest"),
        ("C4 bounce any", t3, o3, bounce, "any"), ("C2 closest", t2, o2, rays2, "closest"), ("C2 any", t2, o2, rays2, "any")]
for name, t, o, r, mode in work:
    t0 = time.time(); w = o.trace(r, mode=mode, nthreads=nt); dt = time.time() - t0
    for k in (0, 1, 2, 3, 4, 5, 6):
        t.set_option("kernel", k)
        assert_hits_equal(t.trace(r, mode=mode), w, f"{name} kernel {k}")
    # device-resident repeats: launches 1-17 of the same batch on one stream -- fresh slot (1), recording launches (2-4, then the host's cadence:
    # 8, 15), orders rebuilt behind them (the rebuild pair in front of launches 2-6, 9, 16), orders reused in between -- for the kernels
    # that claim in a learned order, entry cull on and off, 16-bit and 32-bit lane stacks (round 5: the batch's slot is worked out inside
    # the launch, the header is written by wave 0 of workgroup 0 at the START of the launch: order_commit)
    d_r = torch.from_numpy(r.view(np.uint8).reshape(-1)).cuda()
    d_h = torch.empty(len(r) * 32, dtype=torch.uint8, device="cuda")
    shape = 0
    for s16 in (1, 0):
        t.set_option("stack16", s16)
        for cull in (1, 0):
            t.set_option("entry_cull", cull if mode == "closest" else 2 * cull)
            for k in (3, 5, 6, -1):
                t.set_option("kernel", k)
                shape += 1
                m = len(r) - 128 * shape   # another launch shape (chunk count) per setting: a history of its own (eight at most: the ninth takes over the oldest)
                for rep in range(17):
                    d_h.zero_()
                    t.trace_device(d_r.data_ptr(), d_h.data_ptr(), m, mode=mode)
                    torch.cuda.synchronize()
                    assert_hits_equal(d_h.cpu().numpy().view(rc.HIT_DT)[:m], w[:m], f"{name} kernel {k} entry_cull {cull} stack16 {s16} launch {rep + 1}")
    t.set_option("stack16", 1)
    t.set_option("entry_cull", 1); t.set_option("kernel", -1)
    assert t.get_option("claim_drift") == 0
    print(f"{name}: {len(r)} rays x 7 kernels identical; launches 1-17 of the batch x 4 kernel settings x entry cull on / off x 16- / 32-bit lane stacks identical (oracle {dt:.1f} s, hit fraction {w['hit'].mean():.3f})", flush=True)
