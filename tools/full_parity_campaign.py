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
    print(f"{name}: {len(r)} rays x 7 kernels identical (oracle {dt:.1f} s, hit fraction {w['hit'].mean():.3f})", flush=True)
