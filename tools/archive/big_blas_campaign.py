"""One-off campaign: a BLAS whose packed node array exceeds 4 GiB (34 M triangles) -- the phased kernels' 32-bit buffer offsets do not
reach it, rc_launch_trace must fall back to the 64-bit-pointer kernel; build and hits are checked against the oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import raycore_jl_amd as rc
from oracle import pyoracle as po
from helpers import assert_hits_equal
sc = rc.scenes
n = int(os.environ.get("RC_BIG_N", 34_000_000))
t0 = time.time(); verts = sc.random_triangles(n, 99, edge=0.004); print("gen", round(time.time() - t0, 1), "s", flush=True)
t = rc.TLAS(0)
t0 = time.time(); t.push(verts); t.sync(); print("device build+sync", round(time.time() - t0, 2), "s; nodes bytes", (2 * n - 1) * 64 / 2**30, "GiB", flush=True)
rays = rc.generate_ray_grid(t, (0.3, 0.2, 1.0), 1100)  # > 2 rays per resident thread: auto picks a phased kernel, which must fall back
for k in (-1, 3, 5, 4, 1, 0):
    t.set_option("kernel", k)
    t0 = time.time(); h = t.trace(rays); dt = time.time() - t0
    print("kernel", k, "trace", len(rays), "rays", round(dt, 3), "s (incl. copies), kernel ms", round(t.last_kernel_ms(), 3), "hits", int(h["hit"].sum()), flush=True)
    if k == -1: got = h
    else: assert h.tobytes() == got.tobytes()
o = po.Scene()
t0 = time.time(); o.add_instance(o.add_blas(verts)); o.build(); print("oracle build", round(time.time() - t0, 1), "s", flush=True)
want = o.trace(rays, nthreads=os.cpu_count())
assert_hits_equal(got, want, "big BLAS")
# spot-check the tree: first / last 1000 nodes and root bounds
st = t.adapt()
print("world bound equal:", np.array_equal(np.r_[t.world_bound().p_min, t.world_bound().p_max], o.world_bound), "hits identical: True")
