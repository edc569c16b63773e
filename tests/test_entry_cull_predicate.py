"""The entry cull's claim, checked on the CPU against the oracle alone (no GPU): WHENEVER the product's test would skip an instance entry, the
reference's own traversal of that instance -- recorded by the oracle, entry by entry -- performs no triangle test.  The product's test
and its per-instance spheres are restated in tests/cull_model.py with the kernel's constants; the rays are aimed at the edges of the
derivation (rc_build.hip, above k_inst_recs).  A sphere 20 % too small must be caught: the property is not vacuous."""
import numpy as np
import pytest

import cull_model as cm
from helpers import build_oracle


def spheres_of(o, a_scale=1.0):
    return cm.instance_spheres(o.instances, o.blas_descs, cm.blas_radii(o.blas_descs, o.blas_prims), a_scale)


def violations(o, rays, spheres, mode="closest"):
    """(entries, entries the test would skip, skipped entries in which the reference tested a triangle)"""
    n_ent = n_skip = bad = 0
    for r in rays:
        inst, ct, lf = o.trace_entries(r, mode)
        tmin = np.float32(0) if mode == "any" else r["tmin"]
        for i, c, l in zip(inst, ct, lf):
            n_ent += 1
            if cm.skip_entry(spheres[int(i)], r["o"], r["d"], tmin, c):
                n_skip += 1
                bad += int(l > 0)
    return n_ent, n_skip, bad


def xform(rot=None, scale=1.0, t=(0, 0, 0)):
    r = np.eye(3) if rot is None else np.asarray(rot, dtype=np.float64)
    sc3 = np.full(3, float(scale)) if np.isscalar(scale) else np.asarray(scale, dtype=np.float64)
    m = np.zeros((3, 4)); m[:, :3] = r @ np.diag(sc3); m[:, 3] = t
    return m.astype(np.float32).reshape(12)


def grazing_rays(sc, centres, radii, seed, per_instance):
    g = sc.rng(seed)
    fs = np.array([0.0, 0.5, 0.9, 0.99, 1.0, 1.01, 1.05, 1.08, 1.09, 1.1, 1.11, 1.12, 1.15, 1.2, 1.5, 3.0])
    org, dirs = [], []
    for c, r in zip(centres, radii):
        for _ in range(per_instance):
            d = g.normal(size=3); d /= np.linalg.norm(d)
            u = np.cross(d, g.normal(size=3)); u /= np.linalg.norm(u)
            p = np.asarray(c) + u * r * g.choice(fs) * (1 + g.choice([0, 1e-6, -1e-6, 1e-4, -1e-4]))
            org.append(p - d * g.choice([0.0, 0.3, 3.0, 50.0, 3000.0]) * g.choice([1, 1, 1, -1])); dirs.append(d)
    return sc.make_rays(np.array(org), np.array(dirs))


def test_sphere_lattice_and_a_mutant(oracle):
    import raycore_jl_amd as rc
    sc = rc.scenes
    cfg = sc.config_c3(lon=24, bands=13, lattice=(3, 3, 2))
    o = build_oracle(oracle, cfg)
    xf = np.asarray(cfg["instances"][0][1], dtype=np.float64).reshape(-1, 3, 4)
    rays = grazing_rays(sc, xf[:, :, 3], [0.5 * np.linalg.svd(x[:, :3], compute_uv=False).max() for x in xf], 21, 120)
    rays["tmax"][::3] = sc.rng(5).uniform(0.0, 60.0, len(rays["tmax"][::3]))
    rays = np.concatenate([rays, sc.c3_primary_rays(cfg, 48, 48)])
    for mode in ("closest", "any"):
        n_ent, n_skip, bad = violations(o, rays, spheres_of(o), mode)
        assert bad == 0, (mode, n_ent, n_skip, bad)
        assert n_skip > 0.3 * n_ent, (mode, n_ent, n_skip)             # the cull does its job on this scene
    # teeth: with spheres 20 % too small the same rays find entries that WOULD be skipped although the reference tests triangles there
    assert violations(o, rays, spheres_of(o, a_scale=0.8))[2] > 0


def test_plates_coplanar_rays_and_single_triangle_blas(oracle):
    import raycore_jl_amd as rc
    sc = rc.scenes
    plate = np.array([[0, 0, 0, 1, 0, 0, 1, 1, 0], [0, 0, 0, 1, 1, 0, 0, 1, 0]], dtype=np.float32) - np.float32([0.5, 0.5, 0] * 3)
    four = np.concatenate([plate, plate + np.float32([0, 0, 0.25] * 3)])
    c45 = np.float64(np.float32(np.sqrt(0.5)))
    rz45 = np.array([[c45, -c45, 0], [c45, c45, 0], [0, 0, 1]])
    xfs = np.stack([xform(t=(0, 0, 0)), xform(t=(0, 0, 0)), xform(t=(3, 0, 0)), xform(scale=2.0, t=(0, 4, 1)), xform(t=(40, 0, 0.25)), xform(scale=0.01, t=(-2, -2, 0))])
    lone = np.stack([xform(t=(0, -3, 1.5)), xform(rz45, 1.0, (-4, 3, 1.5)), xform(rz45, 2.0, (8, -3, 0.25))])
    cfg = {"blas": [(four, None), (plate[:1].copy(), None)],
           "instances": [(1, xfs, np.arange(len(xfs), dtype=np.uint32)), (2, lone, np.array([90, 91, 92], dtype=np.uint32))]}
    o = build_oracle(oracle, cfg)
    org, dirs = [], []
    for z in (0.0, 0.25, 1.0, 1.5, 0.125):
        for y in (-0.5, -0.5 - 1e-4, 0.0, 0.5 + 1e-5, 0.7, 4.0, 3.0, -3.0, -3.4, 3.65, 2.35, -4.3):
            for x0 in (-50.0, -5.0, -0.6, 0.0, 2.0, -4.65, 9.3):
                for e1 in (0.0, 1e-7, 9.9e-6, 1e-5, -1.1e-5, 1e-4):
                    for e2 in (0.0, -1e-5, 2e-5):
                        org.append((x0, y, z)); dirs.append((1.0, e1, e2))
                        org.append((y, x0, z)); dirs.append((e1, 1.0, e2))
    rays = sc.make_rays(np.array(org, dtype=np.float64), np.array(dirs, dtype=np.float64))
    sph = spheres_of(o)
    assert all(np.isinf(sph[k][1]) for k in (6, 7, 8)), "single-triangle BLASes are never culled"
    n_ent, n_skip, bad = violations(o, rays, sph)
    assert bad == 0 and n_skip > 0, (n_ent, n_skip, bad)
    # without the single-triangle guard the coplanar rays beside the turned plates would be spared an entry that yields the reference's NaN hit
    radii = cm.blas_radii(o.blas_descs, o.blas_prims)
    unguarded = cm.instance_spheres(o.instances, o.blas_descs, [(c, r, max(n, 2)) for c, r, n in radii])
    assert violations(o, rays, unguarded)[2] > 0


def test_regime_edges_and_hostile_transforms(oracle):
    import raycore_jl_amd as rc
    sc = rc.scenes
    g = sc.rng(78)
    sphere = sc.fan_sphere(16, 9, centre=(0, 0, 0), radius=0.5)
    q, _ = np.linalg.qr(g.normal(size=(3, 3)))
    xfs = np.stack([xform(q, 1.0, (0, 0, 0)), xform(q, 99.0, (300, 0, 0)), xform(q, 101.0, (-400, 0, 0)), xform(q, (1.0, 1.0, 1 / 15.9), (0, 5, 0)),
                    xform(q, (1.0, 1.0, 1 / 16.5), (0, -5, 0)), xform(q, (1.0, 1.0, 0.0), (5, 5, 0)), xform(q, 1e6, (0, 0, 3e6)), xform(q, 1e-6, (1, 1, 1)),
                    xform(q @ np.diag([-1.0, 1, 1]), 0.7, (-3, 2, 1)), xform(q, (0.2, 3.0, 1.0), (2, -3, 2)), xform(None, 1.0, (1e5, 1e5, 0))])
    away = (sphere.reshape(-1, 3) + np.float32([1e4, -2e3, 0])).reshape(-1, 9).astype(np.float32)
    xfs_away = np.stack([xform(None, 1.0, (-1e4, 2e3 + 8, 0)), xform(q, 2.0, tuple(-2.0 * (q @ np.float64([1e4, -2e3, 0])) + np.float64([0, -8, 3])))])
    cfg = {"blas": [(sphere, None), (away, None)],
           "instances": [(1, xfs, np.arange(len(xfs), dtype=np.uint32)), (2, xfs_away, np.array([50, 51], dtype=np.uint32))]}
    o = build_oracle(oracle, cfg)
    centres = np.concatenate([np.asarray(xfs, dtype=np.float64).reshape(-1, 3, 4)[:, :, 3], np.float64([[0, 8, 0], [0, -8, 3]])])
    radii = [0.5 * np.linalg.svd(np.asarray(x, dtype=np.float64).reshape(3, 4)[:, :3], compute_uv=False).max() for x in xfs] + [0.5, 1.0]
    rays = grazing_rays(sc, centres, radii, 13, 90)
    k = np.arange(len(rays))
    for sel, f in ((k % 11 == 0, 0.1), (k % 11 == 1, 0.0999), (k % 11 == 2, 1000.0), (k % 11 == 3, 1001.0), (k % 11 == 4, 1e-7)):
        rays["d"][sel] *= np.float32(f)
    rays["o"][k % 13 == 6] -= rays["d"][k % 13 == 6] * np.float32(2e4)
    rays["o"][k % 71 == 2] += np.float32(3e6)
    sph = spheres_of(o)
    inf = [bool(np.isinf(s[1])) for s in sph]
    assert all(inf[k] for k in (2, 4, 5, 6)), inf          # stretch 101, condition 16.5, singular, stretch 1e6: outside the regime, never culled
    assert not any(inf[k] for k in (0, 1, 7, 8, 10, 11, 12)), inf   # uniform scales 1 / 99 / 1e-6, a mirror, far translations, the far-origin mesh: culled when missed
    n_ent, n_skip, bad = violations(o, rays, sph)
    assert bad == 0 and n_skip > 0, (n_ent, n_skip, bad)


@pytest.mark.parametrize("seed", range(4))
def test_random_blobs(oracle, seed):
    import raycore_jl_amd as rc
    sc = rc.scenes
    g = sc.rng(950 + seed)
    blas = [(sc.random_triangles(int(g.choice([2, 5, 60, 400])), 37 * seed + b, lo=-0.5, hi=0.5, edge=float(g.choice([0.02, 0.3, 1.5]))), None) for b in range(3)]
    instances = []
    for b in range(3):
        m = int(g.integers(2, 7))
        rots = [np.linalg.qr(g.normal(size=(3, 3)))[0] for _ in range(m)]
        xfs = np.stack([xform(r, float(g.uniform(0.2, 4.0)) if g.random() < 0.7 else tuple(g.uniform(0.3, 3.0, 3)), g.uniform(-6, 6, 3)) for r in rots])
        instances.append((b + 1, xfs, g.integers(0, 50, m).astype(np.uint32)))
    o = build_oracle(oracle, {"blas": blas, "instances": instances})
    nr = 2500
    org = g.uniform(-9, 9, size=(nr, 3)); tgt = g.uniform(-7, 7, size=(nr, 3))
    d = tgt - org; d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = sc.make_rays(org, d)
    rays["tmax"][::4] = g.uniform(0, 12, len(rays["tmax"][::4]))
    for mode in ("closest", "any"):
        n_ent, n_skip, bad = violations(o, rays, spheres_of(o), mode)
        assert bad == 0, (mode, n_ent, n_skip, bad)
