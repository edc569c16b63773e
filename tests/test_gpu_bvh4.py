"""GPU parity for the BVH4 path (rc_bvh4.hip) against the oracle's restatement of src/bvh4.jl: the device collapse must
produce a byte-identical BVHNode4 array, closest_hit4 / any_hit4 bit-identical hit records."""
import numpy as np
import pytest

from helpers import assert_hits_equal

pytestmark = pytest.mark.gpu
INVALID = 0xFFFFFFFF


@pytest.fixture(scope="module")
def rc():
    import raycore_jl_amd
    assert raycore_jl_amd.device_count() > 0, "no GPU visible: the product has no CPU fallback"
    return raycore_jl_amd


def soup(n, seed, scale=0.1):
    g = np.random.default_rng(seed)
    c = g.random((n, 1, 3)).astype(np.float32)
    e = (g.random((n, 3, 3)).astype(np.float32) - 0.5) * np.float32(scale)
    return (c + e).reshape(n, 9)


def oracle_scene(po, verts, meta=None):
    s = po.Scene()
    b = s.add_blas(verts, meta)
    s.add_instance(b)
    return s.build(), b


def rays_for(n, seed):
    g = np.random.default_rng(seed)
    o = (g.random((n, 3)) * 2 - 0.5).astype(np.float32)
    d = g.standard_normal((n, 3)).astype(np.float32)
    r = np.zeros(n, dtype=[("o", "<f4", 3), ("tmin", "<f4"), ("d", "<f4", 3), ("tmax", "<f4")])
    r["o"], r["d"], r["tmax"] = o, d, np.inf
    return r


@pytest.mark.parametrize("n,seed,scale", [(1, 1, 0.5), (2, 2, 0.5), (3, 3, 0.5), (5, 4, 0.5), (64, 5, 0.3), (1000, 6, 0.1), (4096, 7, 0.1), (100000, 8, 0.01)])
def test_collapse_is_byte_identical(rc, oracle, n, seed, scale):
    v = soup(n, seed, scale)
    s, b = oracle_scene(oracle, v)
    blas = rc.build_blas4(v)
    want, got = s.blas4_nodes(b), blas.nodes
    assert blas.num_interior == len(want)
    assert got.tobytes() == want.tobytes()


def test_collapse_with_duplicate_codes_and_degenerates(rc, oracle):
    # many identical centroids => Morton ties => deep, skewed BVH2 (the index tie-break), plus dropped degenerate faces
    v = np.tile(soup(8, 21, 0.2), (40, 1))
    v[5] = 0  # degenerate
    s, b = oracle_scene(oracle, v)
    blas = rc.build_blas4(v)
    assert blas.nodes.tobytes() == s.blas4_nodes(b).tobytes()
    r = rays_for(20000, 22)
    assert_hits_equal(blas.trace(r), s.trace4(b, r, nthreads=4), "dup closest4")


@pytest.mark.parametrize("n,seed,scale,n_rays", [(1, 31, 0.8, 4000), (2, 32, 0.8, 4000), (37, 33, 0.4, 20000), (5000, 34, 0.1, 200000), (100000, 35, 0.02, 300000)])
def test_trace4_bit_exact(rc, oracle, n, seed, scale, n_rays):
    v = soup(n, seed, scale)
    meta = np.arange(100, 100 + n, dtype=np.uint32)
    s, b = oracle_scene(oracle, v, meta)
    blas = rc.build_blas4(v, meta)
    r = rays_for(n_rays, seed + 1)
    r["tmin"][::3] = 0.5  # ignored by closest_hit4 / any_hit4 (:610, :700)
    r["tmax"][::5] = 0.7
    want = s.trace4(b, r, nthreads=8)
    got = blas.trace(r)
    assert want["hit"].sum() > 0
    assert_hits_equal(got, want, "closest4")
    assert_hits_equal(blas.trace(r, mode="any"), s.trace4(b, r, mode="any", nthreads=8), "any4")


def test_trace4_matches_bvh2_distance(rc):
    # same geometry through the instanced BVH2 path with an identity instance: t identical, ids equal except exact ties
    v = soup(20000, 41, 0.05)
    blas = rc.build_blas4(v)
    t = rc.TLAS()
    t.push(v)
    t.sync()
    r = rays_for(100000, 42)
    h4, h2 = blas.trace(r), t.trace(r)
    assert np.array_equal(h4["hit"], h2["hit"]) and np.array_equal(h4["t"].view(np.uint32), h2["t"].view(np.uint32))
    assert (h4["primitive_id"] == h2["primitive_id"]).mean() > 0.999


def test_trace4_weird_rays(rc, oracle):
    v = soup(3000, 51, 0.15)
    s, b = oracle_scene(oracle, v)
    blas = rc.build_blas4(v)
    r = rays_for(4096, 52)
    r["d"][::7, 0] = 0.0
    r["d"][::11, 1] = -0.0
    r["d"][::13] = [0, 0, 1]
    r["d"][5] = [0, 0, 0]
    r["tmax"][::17] = 0.0
    r["tmax"][19] = np.nan
    r["o"][23] = [np.nan, 0, 0]
    r["d"][29] = [np.inf, 1, 0]
    r["o"][31] = [1e30, 1e30, 1e30]
    assert_hits_equal(blas.trace(r), s.trace4(b, r), "weird closest4")
    assert_hits_equal(blas.trace(r, mode="any"), s.trace4(b, r, mode="any"), "weird any4")


def test_api_tuples_and_errors(rc):
    tri = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0]], np.float32)
    blas = rc.build_blas4(tri, [42])
    hit, prim, dist, bary = rc.closest_hit4(blas, rc.Ray((0.25, 0.25, 1.0), (0, 0, -1)))
    assert hit and dist == pytest.approx(1.0) and prim.metadata == 42 and bary[0] == pytest.approx(0.5)
    hit, prim, dist, bary = rc.closest_hit4(blas, rc.Ray((2, 2, 1.0), (0, 0, -1)))
    assert not hit and dist == 0 and np.all(prim.vertices == 0) and np.all(bary == 0)
    hit, prim, dist, bary = rc.any_hit4(blas, rc.Ray((2, 2, 1.0), (0, 0, -1)))
    assert not hit and prim.metadata == 42  # dummy = primitives[1] (:763)
    assert rc.any_hit4(blas, rc.Ray((0.25, 0.25, 1.0), (0, 0, -1)))[0]
    assert blas.root_aabb.p_max[0] == 1.0 and len(blas.primitives) == 1
    # tracing a geometry that has no BLAS4 yet is an error, not a fallback
    t = rc.TLAS()
    t.push(tri)
    import ctypes as C
    from raycore_jl_amd import lib
    hits = np.zeros(1, rc.HIT_DT)
    rays = np.zeros(1, rc.RAY_DT)
    assert lib().rc_trace_closest4(t._h, 0, rays.ctypes.data_as(C.c_void_p), hits.ctypes.data_as(C.c_void_p), 1) != 0
    assert lib().rc_trace_closest4(t._h, 7, rays.ctypes.data_as(C.c_void_p), hits.ctypes.data_as(C.c_void_p), 1) != 0


def test_trace4_device_buffers(rc, oracle):
    import torch
    v = soup(4096, 61, 0.1)
    s, b = oracle_scene(oracle, v)
    blas = rc.build_blas4(v)
    r = rays_for(50000, 62)
    d_r = torch.from_numpy(r.view(np.uint8).reshape(-1)).cuda()
    d_h = torch.zeros(len(r) * 32, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    blas.trace_device(d_r.data_ptr(), d_h.data_ptr(), len(r), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = d_h.cpu().numpy().view(rc.HIT_DT)
    assert_hits_equal(got, s.trace4(b, r, nthreads=4), "device closest4")


def test_captured_trace4_on_a_deep_tree_owns_a_spill_region(rc, oracle):
    """a BVH4 launch captured into a hipGraph must get a stack spill region of its own.  Heavily overlapping triangles
    make every ray descend with all four children pushed at every level, so the 24-entry LDS stack of k_trace4 spills into the region on
    the first descent (depth >= 10 => >= 27 entries); with a null region the replay would fault or corrupt.  A stage launch (which runs
    outside the launch guard) after the capture must neither allocate into nor index the capture slots."""
    import torch
    v = soup(100000, 71, 0.6)
    s, b = oracle_scene(oracle, v)
    blas = rc.build_blas4(v)
    blas._scene.sync()  # (build_blas4 pushed an identity instance; the stage launch below wants the scene synced)
    r = rays_for(1500, 72)
    import ctypes as C
    L = oracle.lib()
    L.rco_max_stack.restype, L.rco_max_stack.argtypes = C.c_int32, [C.c_int]
    L.rco_max_stack(1)
    s.trace4(b, r[:300], nthreads=1)
    assert L.rco_max_stack(1) > 24, "the scene no longer drives the stack past k_trace4's 24 LDS entries: the test would prove nothing"
    want, want_any = s.trace4(b, r, nthreads=8), s.trace4(b, r, mode="any", nthreads=8)
    assert_hits_equal(blas.trace(r), want, "eager closest4 (deep)")
    n = len(r)
    d_r = torch.from_numpy(r.view(np.uint8).reshape(-1)).cuda()
    d_h, d_a, d_sh = (torch.zeros(n * 32, dtype=torch.uint8, device="cuda") for _ in range(3))
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        cs = torch.cuda.current_stream().cuda_stream
        blas.trace_device(d_r.data_ptr(), d_h.data_ptr(), n, stream=cs)
        blas.trace_device(d_r.data_ptr(), d_a.data_ptr(), n, mode="any", stream=cs)
    scene = blas._scene
    assert scene.get_option("release_captures") == 2          # two captured launches, each with its own slot and region
    # an unguarded stage launch right after the capture (it used to find a stale capture slot current)
    scene.shadow_rays_device(d_r.data_ptr(), d_h.data_ptr(), n, np.array([3, 3, 3], np.float32), d_sh.data_ptr(), bias=1e-3, stream=st.cuda_stream)
    st.synchronize()
    for rep in range(3):
        d_h.zero_(); d_a.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert_hits_equal(d_h.cpu().numpy().view(rc.HIT_DT), want, f"captured closest4 replay {rep}")
        assert_hits_equal(d_a.cpu().numpy().view(rc.HIT_DT), want_any, f"captured any4 replay {rep}")
    # eager launches beside the graph still work and do not disturb it
    assert_hits_equal(blas.trace(r), want, "eager after capture")
    g.replay()
    torch.cuda.synchronize()
    assert_hits_equal(d_h.cpu().numpy().view(rc.HIT_DT), want, "replay after eager")
    del g
    scene.set_option("release_captures", 1)
    assert scene.get_option("release_captures") == 0
    d_h.zero_()   # (all misses: BVH4 hit records carry no instance, the TLAS-level stage must not be fed them)
    torch.cuda.synchronize()
    scene.shadow_rays_device(d_r.data_ptr(), d_h.data_ptr(), n, np.array([3, 3, 3], np.float32), d_sh.data_ptr(), bias=1e-3, stream=st.cuda_stream)  # after release: nothing to index
    st.synchronize()
    scene.wait_for_gpu()
