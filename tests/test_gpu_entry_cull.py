"""The entry cull (rc_build.hip k_inst_recs, rc_traverse_core.h switch phase): an instance whose entry-cull sphere the ray's segment
misses is not entered.  It may only ever skip entries in which the reference's traversal would test no triangle, so switching it off
must change nothing -- here on the inputs the derivation worries about: rays grazing the sphere at every margin, rays exactly
axis-parallel and exactly coplanar with axis-aligned geometry (determinant 0: the reference reports NaN hits), direction components
on either side of safe_invdir's 1e-5 clamp, origins far from the scene, ray ranges that end inside / before / behind the instance,
|d| at the edges of the regime the cull accepts, and instance transforms at the edges of the regime it accepts (stretch 100,
condition 16) and beyond (singular, huge, tiny: never culled).  Everything is compared with the oracle bit for bit, cull on and off."""
import numpy as np
import pytest

from helpers import assert_hits_equal, build_oracle, build_product

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rc():
    import raycore_jl_amd
    assert raycore_jl_amd.device_count() > 0, "no GPU visible: the product has no CPU fallback"
    return raycore_jl_amd


def xform(rot=None, scale=1.0, t=(0, 0, 0)):
    """3x4 forward transform (12 floats, rows [r | t]): rot @ diag(scale), translation t."""
    r = np.eye(3) if rot is None else np.asarray(rot, dtype=np.float64)
    sc3 = np.full(3, float(scale)) if np.isscalar(scale) else np.asarray(scale, dtype=np.float64)
    m = np.zeros((3, 4), dtype=np.float64)
    m[:, :3] = r @ np.diag(sc3)
    m[:, 3] = t
    return m.astype(np.float32).reshape(12)


def check(rc, t, o, rays, what):
    want_c, want_a = o.trace(rays, nthreads=8), o.trace(rays, mode="any", nthreads=8)
    for cull in (2, 1, 0):          # 2: any_hit batches are culled too (default 1: closest_hit and the drivers)
        t.set_option("entry_cull", cull)
        for kern in (-1, 3, 5, 6):
            t.set_option("kernel", kern)
            assert_hits_equal(t.trace(rays), want_c, f"{what}: closest, entry_cull={cull}, kernel {kern}")
            got = t.trace(rays, mode="any")
            assert np.array_equal(got["hit"], want_a["hit"]), f"{what}: any, entry_cull={cull}, kernel {kern}"
    t.set_option("kernel", -1)
    t.set_option("entry_cull", 1)
    return want_c


def grazing_rays(rc, centres, radii, seed, per_instance=400):
    """Rays aimed to pass each sphere at distance radius * f for f around 1 (the geometry), around the cull sphere (~1.1) and far."""
    g = rc.scenes.rng(seed)
    fs = np.array([0.0, 0.5, 0.9, 0.99, 0.999, 1.0, 1.001, 1.01, 1.05, 1.08, 1.09, 1.1, 1.11, 1.12, 1.15, 1.2, 1.5, 3.0])
    org, dirs = [], []
    for c, r in zip(centres, radii):
        for _ in range(per_instance):
            d = g.normal(size=3); d /= np.linalg.norm(d)
            u = np.cross(d, g.normal(size=3)); u /= np.linalg.norm(u)
            f = g.choice(fs) * (1 + g.choice([0, 1e-6, -1e-6, 1e-4, -1e-4]))
            p = np.asarray(c) + u * r * f                      # closest point of the ray to the centre
            dist = g.choice([0.0, 0.3, 3.0, 50.0, 3000.0])     # origin before / inside / far from the instance
            org.append(p - d * dist * g.choice([1, 1, 1, -1])); dirs.append(d)
    return rc.scenes.make_rays(np.array(org), np.array(dirs))


def test_grazing_a_lattice_of_spheres(rc, oracle):
    sc = rc.scenes
    cfg = sc.config_c3(lon=24, bands=13, lattice=(3, 3, 2))
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    xf = np.asarray(cfg["instances"][0][1], dtype=np.float64).reshape(-1, 3, 4)
    centres = xf[:, :, 3]
    radii = [0.5 * np.linalg.svd(x[:, :3], compute_uv=False).max() for x in xf]
    rays = grazing_rays(rc, centres, radii, 11)
    rays["tmax"][::3] = rc.scenes.rng(5).uniform(0.0, 60.0, len(rays["tmax"][::3]))     # segments that end before / inside / behind
    rays["tmin"][::4] = rc.scenes.rng(6).uniform(-5.0, 5.0, len(rays["tmin"][::4]))
    want = check(rc, t, o, rays, "sphere lattice")
    assert 0.2 < want["hit"].mean() < 0.9
    # the cull does skip entries here (dev counters of the STATS kernel)
    t.set_option("stats", 1); t.set_option("kernel", 5)
    t.trace(rays)
    assert t.get_option("stat19") > 0
    t.set_option("stats", 0); t.set_option("kernel", -1)


def test_axis_aligned_plates_and_coplanar_rays(rc, oracle):
    """Unit plates in the planes z = const (two triangles, several copies stacked exactly: every hit ties), rays exactly in those planes
    (det == 0, u = v = t = NaN: reported as hits by the reference), parallel beside them at offsets around the 1e-5 clamp's reach, and
    with direction components just below / at / above the clamp."""
    sc = rc.scenes
    plate = np.array([[0, 0, 0, 1, 0, 0, 1, 1, 0], [0, 0, 0, 1, 1, 0, 0, 1, 0]], dtype=np.float32) - np.float32([0.5, 0.5, 0] * 3)
    four = np.concatenate([plate, plate + np.float32([0, 0, 0.25] * 3)])             # a BLAS of 4 triangles in two planes
    xfs = np.stack([xform(t=(0, 0, 0)), xform(t=(0, 0, 0)), xform(t=(3, 0, 0)), xform(scale=2.0, t=(0, 4, 1)), xform(t=(40, 0, 0.25)),
                    xform(scale=0.01, t=(-2, -2, 0))])
    # a BLAS of ONE triangle is a single leaf: the reference tests it without any box test (:1553-1570 of the build, :1946 of the loop), so
    # a coplanar ray anywhere inside the TLAS leaf's box gets the NaN hit -- such instances must never be culled
    c45 = np.float64(np.float32(np.sqrt(0.5)))
    rz45 = np.array([[c45, -c45, 0], [c45, c45, 0], [0, 0, 1]])       # a turn about z keeps the plane z = const exact; the world AABB of the
    lone = np.stack([xform(t=(0, -3, 1.5)), xform(scale=3.0, t=(6, 3, 0.125)), xform(rz45, 1.0, (-4, 3, 1.5)), xform(rz45, 2.0, (8, -3, 0.25))])  # turned box outgrows the sphere
    cfg = {"blas": [(four, None), (plate[:1].copy(), None)],
           "instances": [(1, xfs, np.arange(len(xfs), dtype=np.uint32)), (2, lone, np.array([90, 91, 92, 93], dtype=np.uint32))]}
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    org, dirs = [], []
    eps = [0.0, 1e-7, -1e-7, 1e-6, 9.9e-6, 1e-5, 1.1e-5, -1.1e-5, 1e-4, 1e-3]
    for z in (0.0, 0.25, 1.0, 1.5, 0.125):                         # in the planes of the plates and between them
        for y in (-0.5, -0.5 - 1e-4, -0.5 - 1e-6, 0.0, 0.5, 0.5 + 1e-5, 0.7, 4.0, 3.0, 5.0 + 1e-4, -3.0, -3.4, 2.0, 4.4, 3.65, 2.35, -4.3, -1.7):
            for x0 in (-50.0, -5.0, -0.6, 0.0, 2.0, -4.65, -3.35, 9.3, 6.7):
                for e1 in eps:
                    for e2 in (0.0, 1e-6, -1e-5, 2e-5):
                        org.append((x0, y, z)); dirs.append((1.0, e1, e2))
                        org.append((y, x0, z)); dirs.append((e1, 1.0, e2))
    rays = sc.make_rays(np.array(org, dtype=np.float64), np.array(dirs, dtype=np.float64))   # not normalised on purpose: |d|^2 = 1 + O(1e-6)
    want = check(rc, t, o, rays, "plates")
    assert np.isnan(want["t"][want["hit"] == 1]).any(), "the coplanar rays should have produced NaN hits"


def test_regime_edges_of_rays_and_transforms(rc, oracle):
    """|d|^2 at 1e-2 and 1e6 (the cull's ray regime), non-finite and zero directions, origins 1e4..1e7 away; instance stretch at 100 and
    beyond, condition at 16 and beyond, singular / huge / tiny / mirrored / sheared transforms."""
    sc = rc.scenes
    g = sc.rng(77)
    sphere = sc.fan_sphere(16, 9, centre=(0, 0, 0), radius=0.5)
    q, _ = np.linalg.qr(g.normal(size=(3, 3)))
    xfs = np.stack([xform(q, 1.0, (0, 0, 0)), xform(q, 99.0, (300, 0, 0)), xform(q, 101.0, (-400, 0, 0)), xform(q, (1.0, 1.0, 1 / 15.9), (0, 5, 0)),
                    xform(q, (1.0, 1.0, 1 / 16.5), (0, -5, 0)), xform(q, (1.0, 1.0, 0.0), (5, 5, 0)), xform(q, 1e6, (0, 0, 3e6)), xform(q, 1e-6, (1, 1, 1)),
                    xform(q @ np.diag([-1.0, 1, 1]), 0.7, (-3, 2, 1)), xform(q, (0.2, 3.0, 1.0), (2, -3, 2)), xform(None, 1.0, (1e5, 1e5, 0))])
    # a mesh far from its own origin (local coordinates ~1e4: that is where the slab test rounds), brought back by the instance
    away = (sphere.reshape(-1, 3) + np.float32([1e4, -2e3, 0])).reshape(-1, 9).astype(np.float32)
    xfs_away = np.stack([xform(None, 1.0, (-1e4, 2e3 + 8, 0)), xform(q, 2.0, tuple(-2.0 * (q @ np.float64([1e4, -2e3, 0])) + np.float64([0, -8, 3])))])
    cfg = {"blas": [(sphere, None), (away, None)],
           "instances": [(1, xfs, np.arange(len(xfs), dtype=np.uint32)), (2, xfs_away, np.array([50, 51], dtype=np.uint32))]}
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    centres = np.concatenate([np.asarray(xfs, dtype=np.float64).reshape(-1, 3, 4)[:, :, 3], np.float64([[0, 8, 0], [0, -8, 3]])])
    radii = [0.5 * np.linalg.svd(np.asarray(x, dtype=np.float64).reshape(3, 4)[:, :3], compute_uv=False).max() for x in xfs] + [0.5, 1.0]
    rays = grazing_rays(rc, centres, radii, 12, per_instance=300)
    n = len(rays)
    k = np.arange(n)
    for sel, f in ((k % 11 == 0, 0.1), (k % 11 == 1, 0.0999), (k % 11 == 2, 1000.0), (k % 11 == 3, 1001.0), (k % 11 == 4, 1e-7), (k % 11 == 5, 1e12)):
        rays["d"][sel] *= np.float32(f)
    rays["d"][k % 53 == 7] = 0.0
    rays["d"][k % 59 == 3, 0] = np.nan
    rays["d"][k % 61 == 5, 1] = np.inf
    rays["o"][k % 67 == 9, 2] = np.nan
    rays["o"][k % 71 == 2] += np.float32(3e6)
    far = k % 13 == 6
    rays["o"][far] = rays["o"][far] - rays["d"][far] * np.float32(2e4)
    check(rc, t, o, rays, "regime edges")


@pytest.mark.parametrize("seed", range(6))
def test_random_scenes_on_and_off(rc, oracle, seed):
    """Random blobs of triangles (leaf boxes reaching well beyond the vertices' own sphere), random well-conditioned transforms, incoherent rays."""
    sc = rc.scenes
    g = sc.rng(900 + seed)
    blas = [(sc.random_triangles(int(g.choice([2, 5, 60, 700])), 31 * seed + b, lo=-0.5, hi=0.5, edge=float(g.choice([0.02, 0.3, 1.5]))), None) for b in range(3)]
    instances = []
    for b in range(3):
        m = int(g.integers(2, 9))
        rots = [np.linalg.qr(g.normal(size=(3, 3)))[0] for _ in range(m)]
        xfs = np.stack([xform(r, float(g.uniform(0.2, 4.0)) if g.random() < 0.7 else tuple(g.uniform(0.3, 3.0, 3)), g.uniform(-6, 6, 3)) for r in rots])
        instances.append((b + 1, xfs, g.integers(0, 50, m).astype(np.uint32)))
    cfg = {"blas": blas, "instances": instances}
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    nr = 40_000
    org = g.uniform(-9, 9, size=(nr, 3)); tgt = g.uniform(-7, 7, size=(nr, 3))
    d = tgt - org; d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = sc.make_rays(org, d)
    rays["tmax"][::4] = g.uniform(0, 12, len(rays["tmax"][::4]))
    check(rc, t, o, rays, f"random scene {seed}")


def test_inverse_supplied_by_the_caller(rc, oracle):
    """rc_add_instances_with_inverse: the traversal applies the caller's inverse, the TLAS box comes from the forward transform -- and the two
    need not agree.  The cull sphere is derived from the inverse (the matrix the ray really goes through), so a ray is spared the entry
    exactly when the reference, going through the same inverse, reaches no triangle: forward boxes that cover the whole scene with
    inverses that put the geometry somewhere else, and the other way round."""
    sc = rc.scenes
    g = sc.rng(4242)
    sphere = sc.fan_sphere(16, 9, centre=(0, 0, 0), radius=0.5)
    fwd, inv = [], []
    for k in range(8):
        q = np.linalg.qr(g.normal(size=(3, 3)))[0]
        f = np.zeros((3, 4)); f[:, :3] = q * (6.0 if k % 2 else 1.0); f[:, 3] = g.uniform(-3, 3, 3) * (0 if k % 2 else 1)   # odd k: a huge forward box
        where = g.uniform(-3, 3, 3)                                   # where the inverse says the (unit-scale, rotated) geometry is
        q2 = np.linalg.qr(g.normal(size=(3, 3)))[0]
        i = np.zeros((3, 4)); i[:, :3] = q2.T; i[:, 3] = -(q2.T @ where)
        fwd.append(f.astype(np.float32).reshape(12)); inv.append(i.astype(np.float32).reshape(12))
    fwd, inv = np.stack(fwd), np.stack(inv)
    t = rc.TLAS(0)
    t.add_geometry(sphere, None)
    t.push_instances(1, fwd, np.arange(8, dtype=np.uint32), inv_transforms=inv)
    t.sync()
    o = oracle.Scene()
    o.add_blas(sphere, None)
    for k in range(8):
        o.add_instance(1, fwd[k], k, inv=inv[k])
    o.build()
    nr = 60_000
    org = g.uniform(-8, 8, size=(nr, 3)); tgt = g.uniform(-3.5, 3.5, size=(nr, 3))
    d = tgt - org; d /= np.linalg.norm(d, axis=1, keepdims=True)
    want = check(rc, t, o, sc.make_rays(org, d), "caller-supplied inverses")
    assert want["hit"].mean() > 0.02


def test_device_spheres_match_the_model(rc, oracle):
    """tests/cull_model.py restates k_cull_radius / k_inst_recs in numpy; tests/test_entry_cull_predicate.py (CPU) holds that model against
    what the reference does inside every instance.  Here: the spheres the DEVICE computed are the model's."""
    import ctypes
    import cull_model as cm
    sc = rc.scenes
    g = sc.rng(31)
    sphere = sc.fan_sphere(16, 9, centre=(0, 0, 0), radius=0.5)
    blob = sc.random_triangles(300, 5, lo=-0.5, hi=0.5, edge=0.4)
    one = sc.random_triangles(1, 6, lo=-0.5, hi=0.5, edge=0.4)
    away = (sphere.reshape(-1, 3) + np.float32([1e4, -2e3, 0])).reshape(-1, 9).astype(np.float32)
    q = np.linalg.qr(g.normal(size=(3, 3)))[0]
    xfs = [xform(q, 1.0, (0, 0, 0)), xform(q, 99.0, (300, 0, 0)), xform(q, 101.0, (-400, 0, 0)), xform(q, (1.0, 1.0, 1 / 16.5), (0, -5, 0)),
           xform(q, (1.0, 1.0, 0.0), (5, 5, 0)), xform(q, 1e-6, (1, 1, 1)), xform(q @ np.diag([-1.0, 1, 1]), 0.7, (-3, 2, 1)), xform(q, (0.4, 2.0, 1.0), (2, -3, 2))]
    cfg = {"blas": [(sphere, None), (blob, None), (one, None), (away, None)],
           "instances": [(b + 1, np.stack(xfs), np.arange(len(xfs), dtype=np.uint32)) for b in range(4)]}
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    n = len(o.instances)
    dev = np.zeros((n, 8), np.float32)
    hip = ctypes.CDLL("libamdhip64.so")
    t.wait_for_gpu()
    assert hip.hipMemcpy(ctypes.c_void_p(dev.ctypes.data), ctypes.c_void_p(t.get_option("debug_inst_cull_ptr")), dev.nbytes, 2) == 0
    model = cm.instance_spheres(o.instances, o.blas_descs, cm.blas_radii(o.blas_descs, o.blas_prims))
    for i, (cw, A, B) in enumerate(model):
        assert np.isinf(A) == np.isinf(dev[i, 3]), (i, A, dev[i, 3])
        if not np.isinf(A):
            assert np.allclose(dev[i, :3], cw, rtol=2e-6, atol=1e-6 * (1 + np.abs(cw).max())), (i, dev[i, :3], cw)
            assert abs(dev[i, 3] - A) <= 2e-6 * A and abs(dev[i, 4] - B) <= 2e-6 * B, (i, dev[i, 3:5], A, B)
    assert np.isinf(dev[2 * len(xfs):3 * len(xfs), 3]).all()        # the single-triangle BLAS: every instance of it
