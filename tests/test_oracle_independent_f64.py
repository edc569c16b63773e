"""An independent numerical check of the oracle (round-1 verdict, weak #1): every other correctness artefact -- the C oracle, its
brute force, the GPU kernels -- shares ONE formulation of the ray/triangle test (the reference's Moeller-Trumbore in f32).  Here
the hits are recomputed in float64 with a DIFFERENT formulation -- ray/plane intersection, then barycentrics from the edge cross
products -- over all (instance, triangle) pairs, no BVH, and compared with the oracle's closest_hit on BASELINE C1, a C2 sample
and an instanced scene: same hit / miss and the same (instance, primitive) wherever the f64 answer is not within a whisker of a
triangle edge or of a second surface, and t, u, v within the 1e-5 relative tolerance BASELINE.json's north_star states
(src/instanced-bvh.jl:1756-1797 is what is being checked).  A shared restatement error in the dot-product order, the
near/far rule, the transforms or the t-range tests would show up here as a wrong or missing hit."""
import numpy as np
import pytest

from helpers import build_oracle


def f64_closest(verts, prim_meta_unused, instances, rays, chunk=64):
    """verts: (T, 3, 3) f32 local-space triangles of ONE BLAS; instances: list of 3x4 f32 forward transforms.
    Returns per ray: best t, instance, triangle, u, v, and the margin information used to skip ambiguous rays."""
    T = len(verts)
    v0 = verts[:, 0].astype(np.float64); e1 = verts[:, 1].astype(np.float64) - v0; e2 = verts[:, 2].astype(np.float64) - v0
    nrm = np.cross(e1, e2)                       # plane normal (not normalised)
    nn = np.einsum("ij,ij->i", nrm, nrm)
    R = len(rays)
    best_t = np.full(R, np.inf); best_i = np.full(R, -1); best_k = np.full(R, -1)
    best_u = np.zeros(R); best_v = np.zeros(R)
    second_t = np.full(R, np.inf)                # closest OTHER accepted candidate
    edge_margin = np.full(R, np.inf)             # distance of the winning hit's barycentrics from the triangle boundary
    almost_t = np.full(R, np.inf)                # closest triangle that was only just missed (barycentrics within 1e-4 outside)
    for ii, m in enumerate(instances):
        m = np.asarray(m, np.float64).reshape(3, 4)
        A = m[:, :3]; tr = m[:, 3]
        Ainv = np.linalg.inv(A)                  # f64 inverse of the forward transform (the oracle uses the f32 mat3x4_inverse)
        for b in range(0, R, chunk):
            r = rays[b:b + chunk]
            o = (r["o"].astype(np.float64) - tr) @ Ainv.T
            d = r["d"].astype(np.float64) @ Ainv.T
            tmin = r["tmin"].astype(np.float64)[:, None]; tmax = r["tmax"].astype(np.float64)[:, None]
            denom = d @ nrm.T                                            # (r, T)
            with np.errstate(divide="ignore", invalid="ignore"):
                t = np.einsum("rtk,tk->rt", v0[None] - o[:, None], nrm) / denom
                p = o[:, None, :] + t[..., None] * d[:, None, :] - v0[None]   # hit point relative to v0
                u = np.einsum("rtk,tk->rt", np.cross(p, e2[None]), nrm) / nn
                v = np.einsum("rtk,tk->rt", np.cross(e1[None], p), nrm) / nn
            inside = (u >= 0) & (v >= 0) & (u + v <= 1) & (t >= tmin) & (t <= tmax) & np.isfinite(t)
            margin = np.minimum(np.minimum(u, v), 1 - u - v)
            almost = (margin > -1e-4) & (margin < 0) & (t >= tmin) & (t <= tmax) & np.isfinite(t)
            tt = np.where(inside, t, np.inf)
            k = np.argmin(tt, axis=1)
            rows = np.arange(len(r))
            tk = tt[rows, k]
            # second best within this instance
            tt2 = tt.copy(); tt2[rows, k] = np.inf
            t2 = tt2.min(axis=1)
            ta = np.where(almost, t, np.inf).min(axis=1)
            sl = slice(b, b + len(r))
            better = tk < best_t[sl]
            second_t[sl] = np.where(better, np.minimum(best_t[sl], t2), np.minimum(second_t[sl], tk))
            best_u[sl] = np.where(better, u[rows, k], best_u[sl]); best_v[sl] = np.where(better, v[rows, k], best_v[sl])
            edge_margin[sl] = np.where(better, margin[rows, k], edge_margin[sl])
            best_i[sl] = np.where(better, ii, best_i[sl]); best_k[sl] = np.where(better, k, best_k[sl])
            best_t[sl] = np.where(better, tk, best_t[sl])
            almost_t[sl] = np.minimum(almost_t[sl], ta)
    near_miss = almost_t < best_t * (1 + 1e-6)   # some triangle was only just missed in front of (or at) the winner
    return best_t, best_i, best_k, best_u, best_v, second_t, edge_margin, near_miss


def check_scene(po, cfg, rays, tol_uv, what):
    o = build_oracle(po, cfg)
    (verts, _), = cfg["blas"]
    prims = o.blas_prims                              # Morton-sorted order: primitive_id indexes this array
    tri = prims["v"].astype(np.float32)
    xforms = [x for _, xf, _ in cfg["instances"] for x in xf]
    hits = o.trace(rays, nthreads=8)
    t64, i64, k64, u64, v64, t2, margin, near_miss = f64_closest(tri, None, xforms, rays)
    hit64 = np.isfinite(t64)
    # rays whose f64 answer is unambiguous: comfortably inside the triangle, no second surface at nearly the same distance,
    # no just-missed triangle in front
    with np.errstate(invalid="ignore"):
        clear = hit64 & (margin > 1e-4) & ((t2 - t64) > 1e-4 * np.maximum(1.0, np.abs(t64))) & ~near_miss
    clear_miss = ~hit64 & ~near_miss
    assert clear.sum() > 0.2 * hit64.sum() > 0, f"{what}: too few unambiguous rays to say anything"
    # 1. same hit / miss, same instance, same primitive
    assert np.all(hits["hit"][clear] == 1), f"{what}: {int((hits['hit'][clear] != 1).sum())} clear f64 hits are oracle misses"
    assert np.all(hits["hit"][clear_miss] == 0), f"{what}: {int((hits['hit'][clear_miss] != 0).sum())} clear f64 misses are oracle hits"
    assert np.array_equal(hits["instance_id"][clear], i64[clear].astype(np.uint32)), what
    n_per = len(tri)
    assert np.array_equal(hits["primitive_id"][clear], k64[clear].astype(np.uint32)), what  # one BLAS: flat id = sorted index
    # 2. t within 1e-5 relative (north_star), barycentrics within tol_uv
    # (a hit much closer than the scene is large is the difference of two coordinates of the scene's magnitude: its absolute error
    # is f32 epsilon of THOSE, so the relative tolerance is taken against max(|t|, a tenth of the scene diagonal))
    wb = o.world_bound
    floor = 0.1 * float(np.linalg.norm(np.asarray(wb[3:], np.float64) - np.asarray(wb[:3], np.float64)))
    dt = np.abs(hits["t"][clear].astype(np.float64) - t64[clear]) / np.maximum(np.abs(t64[clear]), floor)
    du = np.abs(hits["bary_u"][clear].astype(np.float64) - u64[clear])
    dv = np.abs(hits["bary_v"][clear].astype(np.float64) - v64[clear])
    assert dt.max() <= 1e-5, f"{what}: t off by {dt.max():.3g} relative"
    # u, v are ratios of cross products of (o - v0) with the edges: an f32 evaluation carries |o - v0| / edge units of f32 epsilon
    # (a 1 cm triangle seen from 2 m away: 200), so the 1e-5 tolerance is widened by that conditioning, measured per ray
    idx = np.nonzero(clear)[0]
    cond = np.empty(len(idx))
    for j, r in enumerate(idx):
        m = np.asarray(xforms[i64[r]], np.float64).reshape(3, 4)
        ol = np.linalg.solve(m[:, :3], rays["o"][r].astype(np.float64) - m[:, 3])
        v = tri[k64[r]].astype(np.float64)
        cond[j] = np.linalg.norm(ol - v[0]) / min(np.linalg.norm(v[1] - v[0]), np.linalg.norm(v[2] - v[0]), np.linalg.norm(v[2] - v[1]))
    tol = 1e-5 + tol_uv * cond
    worst = float(np.max(np.maximum(du, dv) / tol))
    assert worst <= 1.0, f"{what}: barycentrics off by {worst:.3g} x (1e-5 + {tol_uv:g} x conditioning)"
    print(f"{what}: {int(clear.sum())} clear hits, {int(clear_miss.sum())} clear misses, max t error {dt.max():.2e}, barycentric error {worst:.2f} of tolerance")
    return int(clear.sum()), int(clear_miss.sum()), float(dt.max()), worst


def scenes_module():
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "raycore.jl_amd", "scenes.py")
    spec = importlib.util.spec_from_file_location("rc_scenes_only", path)   # scene generators only: no HIP library needed
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_c1_every_ray_against_f64_plane_intersection(oracle):
    sc = scenes_module()
    cfg = sc.config_c1()
    o = build_oracle(oracle, cfg)
    rays = o.ray_grid(cfg["viewdir"], cfg["grid"])
    n_hit, n_miss, dt, duv = check_scene(oracle, cfg, rays, 4 * 2.0 ** -23, "C1")
    assert n_hit > 1000 and n_miss > 500


def test_c2_sample_against_f64_plane_intersection(oracle):
    """C2's triangles are ~0.01 across and up to ~2 units from the ray origins: u, v carry the conditioning |o - v0| / edge ~ 200 of
    the f32 cross products (check_scene widens their tolerance by it; t stays within 1e-5)."""
    sc = scenes_module()
    cfg = sc.config_c2()
    o = build_oracle(oracle, cfg)
    rays = o.ray_grid(cfg["viewdir"], cfg["grid"])[sc.rng(1).choice(cfg["grid"] ** 2, 600, replace=False)]
    n_hit, n_miss, dt, duv = check_scene(oracle, cfg, rays, 4 * 2.0 ** -23, "C2 sample")
    assert n_hit > 100


def test_instanced_scene_against_f64_plane_intersection(oracle):
    sc = scenes_module()
    xf, _, _ = sc.lattice_transforms(3, 2, 2, 1.4, 5)
    cfg = {"blas": [(sc.fan_sphere(20, 11), None)], "instances": [(1, xf, np.arange(len(xf), dtype=np.uint32))]}
    o = build_oracle(oracle, cfg)
    wb = o.world_bound
    g = sc.rng(3)
    lo, hi = np.asarray(wb[:3], np.float64), np.asarray(wb[3:], np.float64)
    org = g.uniform(lo - 0.5 * (hi - lo), hi + 0.5 * (hi - lo), size=(3000, 3))
    tgt = g.uniform(lo, hi, size=(3000, 3))
    rays = sc.make_rays(org, sc.normalize(tgt - org))
    rays["tmin"][::7] = 0.7
    rays["tmax"][::5] = 3.0
    n_hit, n_miss, dt, duv = check_scene(oracle, cfg, rays, 16 * 2.0 ** -23, "instanced")  # + the f32 inverse transforms (rotation x scale) the local ray goes through
    assert n_hit > 300 and n_miss > 300
