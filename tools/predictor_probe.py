"""Can a first launch be cost-ordered without a previous launch to learn from (VERDICT r3 #5b)?  Predictor under test: for ONE ray per
128-ray chunk (the middle one), the number of boxes it passes among the top of the tree -- every TLAS node for an instanced scene, the
breadth-first top D levels of the BLAS for a single-instance one -- tested against the segment [t_min, t_max] only (no closest-t pruning:
nothing has been traced yet).  Reported: rank correlation with the chunk's true cost (its longest ray's node visits, instrumented oracle) and
the launch time with the chunks physically reordered by the predictor, by the true cost (the perfect predictor) and in natural order."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raycore_jl_amd as rc
from oracle import pyoracle as po
from tools.perf_probe import build, to_dev
from tools.lpt_probe import oracle_of, timed


def slab_hit(lo, hi, o, inv, tmin, tmax):
    with np.errstate(all="ignore"):
        t0, t1 = (lo - o) * inv, (hi - o) * inv
        tn = np.maximum(np.minimum(t0, t1).max(axis=1), tmin)
        tf = np.minimum(np.maximum(t0, t1).min(axis=1), tmax)
    return tn <= tf


def footprint(nodes, n_leaves, rays, depth, inst_xf=None):
    """per ray: boxes passed among the nodes within `depth` levels of the root (children boxes of the visited interior nodes)"""
    o, d = rays["o"].astype(np.float64), rays["d"].astype(np.float64)
    if inst_xf is not None:  # single instance: into its local frame
        m = inst_xf.astype(np.float64).reshape(3, 4)
        o, d = o @ m[:, :3].T + m[:, 3], d @ m[:, :3].T
    inv = 1.0 / np.where(np.abs(d) > 1e-5, d, np.copysign(1e-5, d))
    tmin, tmax = rays["tmin"].astype(np.float64), rays["tmax"].astype(np.float64)
    count = np.zeros(len(rays), np.int32)
    level = [(1, np.ones(len(rays), bool))]
    for _ in range(depth):
        nxt = []
        for idx, mask in level:
            nd = nodes[idx - 1]
            if nd["child0"] == 0xFFFFFFFF or not mask.any():
                continue
            for c, lo, hi in ((int(nd["child0"]), nd["aabb0_min"], nd["aabb0_max"]), (int(nd["child1"]), nd["aabb1_min"], nd["aabb1_max"])):
                h = mask & slab_hit(lo.astype(np.float64), hi.astype(np.float64), o, inv, tmin, tmax)
                count += h
                if c < n_leaves:
                    nxt.append((c, h))
        level = nxt
    return count


def spearman(a, b):
    ra, rb = np.argsort(np.argsort(a)), np.argsort(np.argsort(b))
    return float(np.corrcoef(ra, rb)[0, 1])


def run(name, t, o, rays, mode, depths):
    _, cnt = o.trace(rays, mode=mode, nthreads=16, counters=True)
    cost = cnt[:, 0].astype(np.int64) + 2 * cnt[:, 1]
    n = len(rays); nc = n // 128
    cmax = cost[:nc * 128].reshape(nc, 128).max(axis=1)
    inst = o.instances
    single = len(inst) == 1
    nodes = o.blas_nodes if single else o.tlas_nodes
    n_leaves = (len(nodes) + 1) // 2
    mid = rays[64:nc * 128:128]
    t.set_option("cost_order", 0)
    def chunks(order):
        return np.concatenate([(order[:, None] * 128 + np.arange(128)[None, :]).reshape(-1), np.arange(nc * 128, n)])
    ref = None
    rows = [("natural", np.arange(nc)), ("true cost (perfect predictor)", np.argsort(-cmax, kind="stable"))]
    for D in depths:
        fp = footprint(nodes, n_leaves, mid, D, inv_of(inst[0]) if single else None)
        q = np.minimum(9, (fp * 10) // (fp.max() + 1))
        rows.append((f"footprint depth {D:2d} (rank corr {spearman(fp, cmax):.3f}), full sort", np.argsort(-fp, kind="stable")))
        rows.append((f"footprint depth {D:2d}, ten classes", np.argsort(-q, kind="stable")))
    print(f"== {name}: {n} rays, {nc} chunks", flush=True)
    for label, order in rows:
        perm = chunks(order)
        ms, hits = timed(t, np.ascontiguousarray(rays[perm]), mode, reps=6)
        back = np.empty_like(hits); back[perm] = hits
        if ref is None: ref = back
        print(f"   {label:70s} {ms:7.3f} ms  {n / ms / 1e3:8.1f} Mrays/s  same hits: {back.tobytes() == ref.tobytes()}", flush=True)
    t.set_option("cost_order", 1)


def inv_of(inst):
    return inst["inv_transform"]


if __name__ == "__main__":
    sc = rc.scenes
    which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["c2", "shadow", "c3_1m", "r1m"]
    if "c2" in which:
        cfg2 = sc.config_c2(); t2 = build(cfg2); o2 = oracle_of(cfg2)
        run("C2 1M coherent", t2, o2, rc.generate_ray_grid(t2, cfg2["viewdir"], cfg2["grid"]), "closest", (5, 7, 9))
    if "shadow" in which or "c3_1m" in which:
        cfg3 = sc.config_c3(); t3 = build(cfg3); o3 = oracle_of(cfg3)
        rays3 = sc.c3_primary_rays(cfg3, 2048, 2048); hits3 = t3.trace(rays3)
        if "shadow" in which:
            run("C3 shadow rays", t3, o3, sc.c3_shadow_rays(cfg3, rays3, hits3), "any", (6, 12))
        if "c3_1m" in which:
            run("C3 primary 1Mi", t3, o3, sc.c3_primary_rays(cfg3, 1024, 1024), "closest", (6, 12))
    if "r1m" in which:
        cfg = {"blas": [(sc.random_triangles(1_000_000, 42, edge=0.01), None)], "instances": [(1, sc.IDENTITY3x4[None], np.zeros(1, np.uint32))]}
        tb = build(cfg); ob = oracle_of(cfg)
        run("random 1M tris, 1M rays", tb, ob, rc.generate_ray_grid(tb, (0.3, 0.2, 1.0), 1000), "closest", (7, 9))
