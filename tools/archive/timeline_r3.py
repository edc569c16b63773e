"""Round 3: per-wave timelines of kernel 5 with and without tapered chunk sizes (option "taper") on the mid-size batches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raycore_jl_amd as rc
from tools.perf_probe import build
from tools.timeline_probe import probe

sc = rc.scenes
cfg2 = sc.config_c2(); t2 = build(cfg2)
rays2 = rc.generate_ray_grid(t2, cfg2["viewdir"], cfg2["grid"])
for tp in (0, 8, 16):
    probe("C2", t2, rays2, opts={"taper": tp})
t2.set_option("taper", 0)
cfg3 = sc.config_c3(); t3 = build(cfg3)
rays3 = sc.c3_primary_rays(cfg3, 2048, 2048)
hits3 = t3.trace(rays3)
shadow = sc.c3_shadow_rays(cfg3, rays3, hits3)
for tp in (0, 8):
    probe("C3 shadow any_hit", t3, shadow, mode="any", opts={"taper": tp})
for tp in (0, 12):
    probe("C3 4Mi", t3, rays3, opts={"taper": tp})
