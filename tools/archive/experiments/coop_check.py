import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import raycore_jl_amd as rc
from oracle import pyoracle as po
from helpers import build_product, build_oracle, assert_hits_equal
sc = rc.scenes
cfg = sc.config_c3(lattice=(3, 3, 2))
t, o = build_product(rc, cfg), build_oracle(po, cfg)
for res in ((8, 8), (64, 48), (320, 200), (700, 500)):
    rays = sc.c3_primary_rays(cfg, *res)
    want = o.trace(rays, nthreads=8)
    t.set_option("kernel", 7)
    got = t.trace(rays)
    assert_hits_equal(got, want, f"coop {res}")
    shadow = sc.c3_shadow_rays(cfg, rays, want)
    assert_hits_equal(t.trace(shadow, mode="any"), o.trace(shadow, mode="any", nthreads=8), f"coop any {res}")
    print("ok", res, len(rays), flush=True)
print("drift", t.get_option("claim_drift"))
