"""Upper bound of cost-ordered claiming (VERDICT r2 #3b): the same C2 / random-geometry / C3-shadow batches with their rays reordered by
the TRUE cost of each ray (node visits counted by the instrumented oracle) -- a perfect predictor, which no pre-pass or previous-frame
estimate can beat -- longest first, either ray by ray or in intact 128-ray chunks (coherent claims), against the natural order.
Results are per-ray properties, so every order must return the same hits (checked)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raycore_jl_amd as rc
from oracle import pyoracle as po
from tools.perf_probe import build, to_dev

def oracle_of(cfg):
    o = po.Scene()
    for v, m in cfg["blas"]: o.add_blas(v, m)
    for b, xf, ids in cfg["instances"]:
        for x, i in zip(xf, ids): o.add_instance(b, x, int(i))
    return o.build()

def timed(t, rays, mode, reps=5):
    d_r, d_h = to_dev(rays), torch.empty(len(rays) * 32, dtype=torch.uint8, device="cuda")
    best = 1e9
    for _ in range(reps):
        t.trace_device(d_r.data_ptr(), d_h.data_ptr(), len(rays), mode=mode)
        best = min(best, t.last_kernel_ms())
    return best, d_h.cpu().numpy().view(rc.HIT_DT)

def run(name, t, o, rays, mode):
    threads = 16
    _, cnt = o.trace(rays, mode=mode, nthreads=threads, counters=True)
    cost = cnt[:, 0].astype(np.int64) + 2 * cnt[:, 1]
    n = len(rays)
    print(f"== {name}: {n} rays, node visits per ray mean {cost.mean():.1f} p50 {np.percentile(cost, 50):.0f} p99 {np.percentile(cost, 99):.0f} max {cost.max()}", flush=True)
    orders = {"natural": np.arange(n)}
    orders["rays by cost, longest first"] = np.argsort(-cost, kind="stable")
    nc = n // 128
    cmax = cost[:nc * 128].reshape(nc, 128).max(axis=1)
    chunk_order = np.argsort(-cmax, kind="stable")
    orders["128-ray chunks by their longest ray, longest first"] = np.concatenate([(chunk_order[:, None] * 128 + np.arange(128)[None, :]).reshape(-1), np.arange(nc * 128, n)])
    csum = cost[:nc * 128].reshape(nc, 128).sum(axis=1)
    chunk_order = np.argsort(-csum, kind="stable")
    orders["128-ray chunks by total work, heaviest first"] = np.concatenate([(chunk_order[:, None] * 128 + np.arange(128)[None, :]).reshape(-1), np.arange(nc * 128, n)])
    q = np.minimum(9, (cmax * 10) // (cmax.max() + 1))           # ten linear classes of the chunk's longest ray, stable inside a class
    chunk_order = np.argsort(-q, kind="stable")
    orders["128-ray chunks in ten classes of their longest ray"] = np.concatenate([(chunk_order[:, None] * 128 + np.arange(128)[None, :]).reshape(-1), np.arange(nc * 128, n)])
    heavy = cmax >= np.percentile(cost, 99)
    chunk_order = np.argsort(~heavy, kind="stable")
    orders["128-ray chunks holding a ray above the 99th percentile first"] = np.concatenate([(chunk_order[:, None] * 128 + np.arange(128)[None, :]).reshape(-1), np.arange(nc * 128, n)])
    # long rays first, but only the top 5 % moved to the front (keeps the rest of the image coherent)
    top = np.argsort(-cost, kind="stable")[: n // 20]
    mask = np.ones(n, bool); mask[top] = False
    orders["longest 5 % of the rays first, the rest in natural order"] = np.concatenate([top, np.nonzero(mask)[0]])
    ref = None
    t.set_option("cost_order", 0)
    for taper in (0, 12):
        t.set_option("taper", taper)
        for label, perm in orders.items():
            ms, hits = timed(t, np.ascontiguousarray(rays[perm]), mode)
            back = np.empty_like(hits); back[perm] = hits
            if ref is None: ref = back
            same = back.tobytes() == ref.tobytes()
            print(f"   taper {taper:2d}  {label:60s} {ms:7.3f} ms  {n / ms / 1e3:8.1f} Mrays/s  same hits: {same}", flush=True)
    t.set_option("taper", 12)
    for label, opts in (("product: taper 12, cost_order off", {"cost_order": 0}), ("product: cost_order on (learned from the previous launch)", {"cost_order": 1, "cost_thr": 64}),
                        ("product: cost_order machinery with nothing to report (identity order: its overhead)", {"cost_order": 1, "cost_thr": 4096})):
        for k, v in opts.items(): t.set_option(k, v)
        ms, hits = timed(t, rays, mode, reps=8)
        print(f"   {label:95s} {ms:7.3f} ms  {n / ms / 1e3:8.1f} Mrays/s  same hits: {hits.tobytes() == ref.tobytes()}", flush=True)
    t.set_option("cost_order", 1); t.set_option("cost_thr", 64)

sc = rc.scenes
cfg2 = sc.config_c2(); t2 = build(cfg2); o2 = oracle_of(cfg2)
run("C2 1M coherent", t2, o2, rc.generate_ray_grid(t2, cfg2["viewdir"], cfg2["grid"]), "closest")
cfg3 = sc.config_c3(); t3 = build(cfg3); o3 = oracle_of(cfg3)
rays3 = sc.c3_primary_rays(cfg3, 2048, 2048); hits3 = t3.trace(rays3)
run("C3 shadow rays", t3, o3, sc.c3_shadow_rays(cfg3, rays3, hits3), "any")
run("C3 primary 1Mi", t3, o3, sc.c3_primary_rays(cfg3, 1024, 1024), "closest")
