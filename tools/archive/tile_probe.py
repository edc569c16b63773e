"""Dev tool: does the order of a coarse ray grid matter?  1000 x 1000 rays (the get_illumination default shape) traced in grid order
(a wave's 128-ray claim = one strip of a row) and in 2-D tile orders (a claim = one compact tile), on C2 and on the reference's
random-geometry benchmark scenes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raycore_jl_amd as rc
from perf_probe import build, time_trace
sc = rc.scenes


def orders(g):
    idx = np.arange(g * g).reshape(g, g)  # [j][i], ray index = i + g*j
    out = {"grid order": idx.reshape(-1)}
    for tw, th in ((16, 8), (8, 16), (32, 4), (8, 8)):
        gw, gh = g // tw * tw, g // th * th
        core = idx[:gh, :gw].reshape(gh // th, th, gw // tw, tw).transpose(0, 2, 1, 3).reshape(-1)
        rest = np.concatenate([idx[:gh, gw:].reshape(-1), idx[gh:, :].reshape(-1)])
        out[f"{tw}x{th} tiles"] = np.concatenate([core, rest])
    # Morton order of (i, j)
    def part(x):
        x = x.astype(np.uint64)
        x = (x | (x << 8)) & 0x00FF00FF
        x = (x | (x << 4)) & 0x0F0F0F0F
        x = (x | (x << 2)) & 0x33333333
        x = (x | (x << 1)) & 0x55555555
        return x
    jj, ii = np.divmod(np.arange(g * g), g)
    out["morton"] = np.argsort(part(ii) | (part(jj) << 1), kind="stable")
    return out


def run(name, t, viewdir, g):
    rays = rc.generate_ray_grid(t, viewdir, g)
    base = None
    for oname, perm in orders(g).items():
        ms, hits = time_trace(t, rays[perm], "closest", 5)
        inv = np.empty_like(perm); inv[perm] = np.arange(len(perm))
        h = hits[inv]
        if base is None:
            base = h
        ok = h.tobytes() == base.tobytes()
        print(f"{name:22s} {oname:14s} {ms:8.3f} ms {len(rays)/ms/1e3:8.1f} Mrays/s {'' if ok else 'RESULTS DIFFER'}", flush=True)


cfg = sc.config_c2()
t2 = build(cfg)
run("C2 100k tris", t2, cfg["viewdir"], 1000)
run("C2 100k tris g=2000", t2, cfg["viewdir"], 2000)
t2.free()
for nt in (250_000, 1_000_000):
    tb = rc.TLAS(0)
    tb.add_geometry(sc.random_triangles(nt, 42, edge=0.01))
    tb.push_instances(1)
    tb.sync()
    run(f"random {nt // 1000}k tris", tb, (0.3, 0.2, 1.0), 1000)
    tb.free()
