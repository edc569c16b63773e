"""Tight-loop lifecycle stress restated from the reference's test/test_tlas_stress.jl (GPU): many refit frames, many rebuild
frames, update / trace interleaving, delete + push without sync, drain to empty.  The reference's StaticTLAS-identity checks
(`tlas.static_tlas === st0`) become "the sync was a refit and the adapted arrays kept their sizes"; its blas_storage /
flat-array leak checks become the geometry / primitive / node counters of rc_counts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rc():
    import raycore_jl_amd
    assert raycore_jl_amd.device_count() > 0
    return raycore_jl_amd


def sphere(rc, n):
    """stress_sphere(n): unit sphere, apex exactly at z = 1 (so a ray down the axis hits at t = z0 - z_off - 1)."""
    return rc.scenes.fan_sphere(2 * n, n, radius=1.0)


def xlats(n, y=0.0, z=0.0):
    m = np.tile(np.array([1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0], np.float32), (n, 1))
    m[:, 3] = np.arange(1, n + 1, dtype=np.float32) * np.float32(0.1)
    m[:, 7] = y
    m[:, 11] = z
    return m


def down_rays(rc, xs, y, z0=5.0):
    o = np.array([[np.float32(i) * np.float32(0.1), y, z0] for i in xs], np.float32)
    return rc.scenes.make_rays(o, np.tile([0, 0, -1], (len(xs), 1)))


def counts(t):
    live, total, geoms, prims, tlas_nodes, blas_nodes = t._counts()
    return live, geoms, prims, blas_nodes, tlas_nodes


def test_5000_instances_200_refit_frames(rc):  # test/test_tlas_stress.jl:284-327
    t = rc.TLAS()
    n = 5000
    h = t.push(sphere(rc, 4), xlats(n))
    t.sync()
    base = counts(t)
    for frame in range(1, 201):
        t.update_transforms(h, xlats(n, y=np.float32(0.1 * frame)))
        t.sync()
        assert t.last_sync_action == "refit" and counts(t) == base and t.n_instances() == n and t.n_geometries() == 1
    hits = t.trace(down_rays(rc, [1, n // 2, n], np.float32(0.1 * 200)))
    assert np.all(hits["hit"] == 1) and np.all(np.abs(hits["t"] - 4.0) < 0.1)


def test_2000_instances_100_rebuild_frames(rc):  # :333-378
    t = rc.TLAS()
    n = 2000
    mesh = sphere(rc, 4)
    h = t.push(mesh, xlats(n))
    t.sync()
    base = counts(t)
    for frame in range(1, 101):
        t.delete(h)
        h = t.push(mesh, xlats(n, y=np.float32(0.05 * frame)))
        t.sync()
        assert t.last_sync_action == "rebuild" and counts(t) == base  # one geometry, no leaked primitives / nodes
    hits = t.trace(down_rays(rc, [1, n // 2, n], np.float32(0.05 * 100)))
    assert np.all(hits["hit"] == 1)


@pytest.mark.parametrize("n,frames,samples", [(1000, 100, (1, 250, 500, 750, 1000)), (5000, 50, (1, 1000, 2500, 4000, 5000))])
def test_interleaved_update_and_trace(rc, n, frames, samples):  # :384-445
    t = rc.TLAS()
    h = t.push(sphere(rc, 4), xlats(n))
    t.sync()
    base = counts(t)
    for frame in range(1, frames + 1):
        z_off = np.float32((frame % 50) * 0.04) if n == 1000 else np.float32(frame * 0.04)
        t.update_transforms(h, xlats(n, z=z_off))
        t.sync()
        assert t.last_sync_action == "refit" and counts(t) == base
        hits = t.trace(down_rays(rc, samples, 0.0))
        assert np.all(hits["hit"] == 1) and np.all(np.abs(hits["t"] - (5.0 - z_off - 1.0)) < 0.1), frame


def test_interleaved_delete_push_sync_trace(rc):  # :451-511
    t = rc.TLAS()
    n = 500
    h = t.push(sphere(rc, 4), xlats(n))
    t.sync()
    for frame in range(1, 61):
        t.delete(h)
        z_off = np.float32((frame % 40) * 0.05)
        mesh = sphere(rc, 4 if frame % 2 else 8)
        h = t.push(mesh, xlats(n, z=z_off))
        t.sync()
        live, geoms, prims, blas_nodes, _ = counts(t)
        assert live == n and geoms == 1 and prims == len(mesh) and blas_nodes == 2 * len(mesh) - 1
        hits = t.trace(down_rays(rc, (1, 100, 250, 400, n), 0.0))
        assert np.all(hits["hit"] == 1) and np.all(np.abs(hits["t"] - (5.0 - z_off - 1.0)) < 0.1), frame


def test_200_swaps_exact_compaction(rc):  # :554-579
    t = rc.TLAS()
    one = np.array([[1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0]], np.float32)
    h = t.push(sphere(rc, 16), one)
    t.sync()
    for it in range(1, 201):
        nn = 64 if it % 5 == 0 else (8 if it % 2 == 0 else 24)
        mesh = sphere(rc, nn)
        t.delete(h)
        x = one.copy()
        x[0, 11] = np.float32(0.001 * it)
        h = t.push(mesh, x)
        t.sync()
        live, geoms, prims, blas_nodes, _ = counts(t)
        assert (live, geoms, prims, blas_nodes) == (1, 1, len(mesh), 2 * len(mesh) - 1)
    hit, _, dist, _, _ = rc.closest_hit(t, rc.Ray((0, 0, 5), (0, 0, -1)))
    assert hit and abs(dist - (5.0 - 0.001 * 200 - 1.0)) < 0.15


def test_500_refit_cycles_keep_the_adapted_form(rc):  # :623-650
    t = rc.TLAS()
    h = t.push(sphere(rc, 16))
    t.sync()
    base = counts(t)
    for it in range(1, 501):
        m = np.eye(4, dtype=np.float32)
        m[2, 3] = np.float32(it * 0.001)
        t.update_transform(h, m)
        t.sync()
        assert t.last_sync_action == "refit" and counts(t) == base
    assert t.sync().last_sync_action == "noop"  # a clean sync does nothing (:898-900)
    hit, _, dist, _, _ = rc.closest_hit(t, rc.Ray((0, 0, 5), (0, 0, -1)))
    assert hit and abs(dist - (5.0 - 0.5 - 1.0)) < 0.1


def test_topology_change_after_long_refit_run(rc):  # :656-687
    def xl(x, y, z):
        m = np.eye(4, dtype=np.float32)
        m[:3, 3] = [x, y, z]
        return m
    t = rc.TLAS()
    ha = t.push(sphere(rc, 8), xl(-2, 0, 0))
    hb = t.push(sphere(rc, 8), xl(2, 0, 0))
    t.sync()
    for it in range(1, 101):
        t.update_transform(ha, xl(-2 + it * 0.01, 0, 0))
        t.update_transform(hb, xl(2 - it * 0.01, 0, 0))
        t.sync()
        assert t.last_sync_action == "refit"
    t.delete(ha)
    hc = t.push(sphere(rc, 8), xl(0, 0, 5))
    t.sync()
    assert t.last_sync_action == "rebuild" and t.n_instances() == 2 and t.n_geometries() == 2
    assert not rc.closest_hit(t, rc.Ray((-2 + 100 * 0.01, 0, 5), (0, 0, -1)))[0]
    hit, _, dist, _, _ = rc.closest_hit(t, rc.Ray((0, 0, 10), (0, 0, -1)))
    assert hit and abs(dist - 4.0) < 0.1 and t.is_valid(hc) and t.is_valid(hb) and not t.is_valid(ha)


def test_delete_and_push_without_intermediate_sync(rc):  # :769-802
    def xl(x):
        m = np.eye(4, dtype=np.float32)
        m[0, 3] = x
        return m
    t = rc.TLAS()
    h1, h2, h3 = (t.push(sphere(rc, 8), xl(x)) for x in (0, 2, 4))
    t.sync()
    assert t.n_instances() == 3
    t.delete(h2)
    h4 = t.push(sphere(rc, 8), xl(6))
    t.delete(h1)
    h5 = t.push(sphere(rc, 8), xl(8))
    t.sync()
    assert all(t.is_valid(h) for h in (h3, h4, h5)) and not t.is_valid(h1) and not t.is_valid(h2)
    assert t.n_instances() == 3 and t.n_geometries() == 3
    o = np.array([[x, 0, 5] for x in (0, 2, 4, 6, 8)], np.float32)
    hits = t.trace(rc.scenes.make_rays(o, np.tile([0, 0, -1], (5, 1))))
    assert list(hits["hit"]) == [0, 0, 1, 1, 1]


def test_drain_to_empty_and_rebuild(rc):  # :808-836
    t = rc.TLAS()
    for it in range(1, 6):
        m = np.eye(4, dtype=np.float32)
        m[0, 3] = it
        h = t.push(sphere(rc, 8), m)
        t.sync()
        assert t.n_instances() == 1 and t.n_geometries() == 1
        t.delete(h)
        t.sync()
        live, geoms, prims, blas_nodes, tlas_nodes = counts(t)
        assert (live, geoms, prims, blas_nodes, tlas_nodes) == (0, 0, 0, 0, 0)
        assert not rc.closest_hit(t, rc.Ray((0, 0, 5), (0, 0, -1)))[0]


def test_adapt_per_dispatch_sees_mutations_and_refit_path(rc):  # test/test_mesh_update.jl:118-223
    def xl(z):
        m = np.eye(4, dtype=np.float32)
        m[2, 3] = z
        return m
    ray = rc.Ray((0, 0, 5), (0, 0, -1))
    t = rc.TLAS()
    h = t.push(sphere(rc, 16), xl(0))
    st = t.adapt()                                   # first adapt builds the adapted form
    hit, _, dist, _, _ = rc.closest_hit(st, ray)
    assert hit and abs(dist - 4.0) < 0.05
    t.delete(h)
    h = t.push(sphere(rc, 48), xl(2.0))              # mutate: the canonical consumer re-adapts per dispatch and must see it
    hit, _, dist, _, _ = rc.closest_hit(t.adapt(), ray)
    assert hit and abs(dist - 2.0) < 0.1
    assert t.sync().last_sync_action == "noop"       # clean sync: no-op (:898-900)
    # refit path: update_transform! + sync! moves the instance in place
    t.update_transform(h, xl(1.5))
    assert t.sync().last_sync_action == "refit"
    hit, _, dist, _, _ = rc.closest_hit(t, ray)
    assert hit and abs(dist - 2.5) < 0.05
    for _ in range(3):
        assert t.sync().last_sync_action == "noop"
