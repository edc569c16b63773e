"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json): hit flag / primitive id / instance id bit-exact; t and barycentrics within 1e-5
relative.  Because the kernels keep the reference's expression order with FMA contraction off, these tests
assert the stronger property that t, u, v are BIT-identical.
"""
import os

import numpy as np
import pytest

from helpers import assert_hits_equal, build_oracle, build_product, random_rays

pytestmark = pytest.mark.gpu


# Tests below that assert the DEFAULT shapes' plane counts / stack widths are skipped when a campaign runs the suite with RC_STACK16=0 (32-bit lane stacks as
# the process default, profiles/r06_parity_campaigns.txt): the parity they check is covered there by the fuzz run under the same setting.
default_stack_shape = pytest.mark.skipif(os.environ.get("RC_STACK16", "1") == "0", reason="asserts the 16-bit-stack shapes; RC_STACK16=0 is set")


@pytest.fixture(scope="module")
def rc():
    import raycore_jl_amd
    assert raycore_jl_amd.device_count() > 0, "no GPU visible: the product has no CPU fallback"
    return raycore_jl_amd


UNIT_TRI = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0]], dtype=np.float32)


def xlat(x, y, z):
    m = np.eye(4, dtype=np.float32)
    m[:3, 3] = [x, y, z]
    return m


def nodes_equal(a, b):
    return a.tobytes() == b.tobytes()


# ---- reference KATs restated against the product (test/test_instanced_bvh.jl, test/test_intersection.jl) ----
def test_kat_closest_hit_basic(rc):  # test/test_instanced_bvh.jl:274-302
    t = rc.TLAS()
    t.push(UNIT_TRI, meta=[42])
    hit, prim, dist, bary, inst = rc.closest_hit(t, rc.Ray((0.25, 0.25, 1.0), (0, 0, -1)))
    assert hit and dist == pytest.approx(1.0) and prim.metadata == 42 and inst == 1
    assert bary[0] == pytest.approx(0.5, abs=0.01) and bary[1] == pytest.approx(0.25, abs=0.01)
    hit, prim, dist, bary, inst = rc.closest_hit(t, rc.Ray((2, 2, 1.0), (0, 0, -1)))
    assert not hit and dist == 0 and inst == 0 and np.all(prim.vertices == 0) and np.all(bary == 0)  # test/test_intersection.jl:120-142


def test_kat_translated_and_nearest(rc):  # test/test_instanced_bvh.jl:304-378
    t = rc.TLAS()
    t.push(UNIT_TRI, xlat(10, 0, 0))
    assert not rc.closest_hit(t, rc.Ray((0.25, 0.25, 1.0), (0, 0, -1)))[0]
    hit, _, dist, _, _ = rc.closest_hit(t, rc.Ray((10.25, 0.25, 1.0), (0, 0, -1)))
    assert hit and dist == pytest.approx(1.0)
    t2 = rc.TLAS()
    t2.push(UNIT_TRI, [xlat(0, 0, 0), xlat(0, 0, -5)], instance_ids=[1, 2])
    hit, _, dist, _, inst = rc.closest_hit(t2, rc.Ray((0.25, 0.25, 1.0), (0, 0, -1)))
    assert hit and dist == pytest.approx(1.0) and inst == 1


def test_kat_any_hit(rc):  # test/test_instanced_bvh.jl:380-405, 847-876
    t = rc.TLAS()
    t.push(UNIT_TRI)
    o = [[0.25, 0.25, 1.0], [0.1, 0.1, 1.0], [5.0, 5.0, 1.0], [0.9, 0.9, 1.0]]
    h = t.sync().trace(rc.scenes.make_rays(o, [0, 0, -1]), mode="any")
    assert list(h["hit"]) == [1, 1, 0, 0]
    assert rc.any_hit(t, rc.Ray((0.25, 0.25, 1.0), (0, 0, -1)))[0]
    miss = rc.any_hit(t, rc.Ray((2, 2, 1.0), (0, 0, -1)))
    assert not miss[0] and np.array_equal(miss[1].vertices.reshape(-1), UNIT_TRI[0])  # dummy = all_blas_prims[1], src/instanced-bvh.jl:2137


def test_kat_instance_ids_and_node_counts(rc):  # test/test_instanced_bvh.jl:788-805, 878-916
    t = rc.TLAS()
    t.push(UNIT_TRI, [xlat(0, 0, 0), xlat(5, 0, 0), xlat(0, 5, 0)])
    o = [[0.25, 0.25, 1.0], [5.25, 0.25, 1.0], [0.25, 5.25, 1.0]]
    h = t.sync().trace(rc.scenes.make_rays(o, [0, 0, -1]))
    assert all(h["hit"] == 1) and list(h["instance_id"] + 1) == [1, 2, 3]
    t81 = rc.TLAS()
    t81.push(UNIT_TRI, [xlat(((i - 1) % 9) * 1.5, ((i - 1) // 9) * 1.25, 0) for i in range(1, 82)])
    st = t81.adapt()
    assert len(st.instances) == 81 and len(st.nodes) == 161


def test_kat_full_trace_and_bary(rc):  # test/test_instanced_bvh.jl:954-1042
    t, _ = rc.TLAS_from_meshes([UNIT_TRI, UNIT_TRI + np.tile([5, 0, 0], 3).astype(np.float32)])
    o = [[0.25, 0.25, 2.0], [5.25, 0.25, 3.0], [10.0, 10.0, 1.0]]
    res = rc.trace_rays(t, rc.scenes.make_rays(o, [0, 0, -1]))
    assert [r[0] for r in res] == [True, True, False]
    assert res[0][2] == pytest.approx(2.0) and res[1][2] == pytest.approx(3.0)
    assert res[0][4] == 1 and res[1][4] == 2
    assert res[0][3][0] == pytest.approx(0.5, abs=0.01) and res[1][3][0] == pytest.approx(0.5, abs=0.01)


def test_kat_tlas_items_ctor(rc):  # test/test_intersection.jl:57-103
    meshes = [np.array([[-1, -1, z, 1, -1, z, 0, 1, z]], dtype=np.float32) for z in (0, 4, 8)]
    accel = rc.TLAS_from_items(meshes, lambda mi, ti: mi)
    hit, tri, dist, _, inst = rc.closest_hit(accel, rc.Ray((0, 0, -2), (0, 0, 1)))
    assert hit and dist == pytest.approx(2.0) and tri.metadata == 1 and inst == 1
    assert accel.instances["instance_id"].tolist() == [1, 2, 3]


def test_kat_dynamic_scenes(rc):  # test/test_instanced_bvh.jl:1044-1132, 1178-1194
    t = rc.TLAS()
    h = t.push(UNIT_TRI)
    t.sync()
    assert rc.closest_hit(t.adapt(), rc.Ray((0.25, 0.25, 1.0), (0, 0, -1)))[0]
    t.update_transform(h, xlat(10, 0, 0))
    t.sync()
    st = t.adapt()
    assert not rc.closest_hit(st, rc.Ray((0.25, 0.25, 1.0), (0, 0, -1)))[0]
    hit, _, dist, _, _ = rc.closest_hit(st, rc.Ray((10.25, 0.25, 1.0), (0, 0, -1)))
    assert hit and dist == pytest.approx(1.0)
    # add an instance of a second mesh between dispatches
    t2 = rc.TLAS()
    t2.push(UNIT_TRI)
    rays = [rc.Ray((0.25, 0.25, 1.0), (0, 0, -1)), rc.Ray((5.25, 0.25, 1.0), (0, 0, -1))]
    assert [r[0] for r in rc.trace_rays(t2.adapt(), rays)] == [True, False]
    t2.push(UNIT_TRI + np.array([5, 0, 0] * 3, np.float32))
    assert [r[0] for r in rc.trace_rays(t2.adapt(), rays)] == [True, True]
    # world bound of the adapted form = the TLAS root box
    t3 = rc.TLAS()
    t3.push(UNIT_TRI, [xlat(0, 0, 0), xlat(10, 10, 0)])
    t3.sync()  # root_aabb is the synced TLAS's (the reference test syncs before reading it)
    wb, root = t3.world_bound(), t3.adapt().root_aabb
    assert np.allclose(wb.p_min, [0, 0, 0]) and np.allclose(wb.p_max, [11, 11, 0]) and np.array_equal(root.p_min, wb.p_min)


def test_empty_tlas_traces_miss(rc):  # test/test_tlas_stress.jl:808-831
    t = rc.TLAS()
    for it in range(3):
        h = t.push(UNIT_TRI, xlat(it, 0, 0))
        t.sync()
        assert t.n_instances() == 1 and t.n_geometries() == 1
        assert t.delete(h) and not t.delete(h)
        t.sync()
        assert t.n_instances() == 0 and t.n_geometries() == 0 and t.n_primitives() == 0
        assert not rc.closest_hit(t, rc.Ray((0, 0, 5), (0, 0, -1)))[0]


# ---- device-built BVH == oracle BVH, byte for byte -------------------------------------------------------
@pytest.mark.parametrize("cfg_name", ["c1", "c3", "random_multi", "flat_dupes"])
def test_build_parity(rc, oracle, cfg_name):
    sc = rc.scenes
    if cfg_name == "c1":
        cfg = sc.config_c1()
    elif cfg_name == "c3":
        cfg = sc.config_c3()
    elif cfg_name == "random_multi":
        xf, _, _ = sc.lattice_transforms(3, 3, 2, 1.2, 77)
        cfg = {"blas": [(sc.random_triangles(5000, 5, edge=0.05), None), (sc.fan_sphere(16, 9), None), (UNIT_TRI, [7])],
               "instances": [(1, xf[:6], np.arange(6, dtype=np.uint32)), (2, xf[6:15], np.arange(9, dtype=np.uint32) + 10),
                             (3, xf[15:], np.zeros(3, np.uint32))]}
    else:  # coplanar geometry (zero extent => NaN Morton axis) with exact duplicates (index tie-break)
        quad = np.array([[0, 0, 0, 1, 0, 0, 1, 1, 0], [0, 0, 0, 1, 1, 0, 0, 1, 0]], dtype=np.float32)
        cfg = {"blas": [(np.concatenate([quad] * 40), None)],
               "instances": [(1, np.stack([sc.IDENTITY3x4] * 5), np.zeros(5, np.uint32))]}
    t = build_product(rc, cfg)
    o = build_oracle(oracle, cfg)
    st = t.adapt()
    assert nodes_equal(st.all_blas_nodes, o.blas_nodes)
    assert nodes_equal(st.nodes, o.tlas_nodes)
    assert st.all_blas_prims.tobytes() == o.blas_prims.tobytes()
    assert st.blas_descriptors.tobytes() == o.blas_descs.tobytes()
    assert st.instances.tobytes() == o.instances.tobytes()
    wb = t.world_bound()
    assert np.array_equal(np.concatenate([wb.p_min, wb.p_max]), o.world_bound)


def test_build_from_device_buffers(rc, oracle):
    """rc_add_blas_device: soup (with degenerate faces and explicit metadata) already in device memory."""
    import torch
    verts = rc.scenes.uv_sphere_grid(40, 30, centre=(1, 2, 3), radius=2.0)   # has exactly-degenerate pole faces
    meta = (np.arange(len(verts), dtype=np.uint32) * 7 + 3)
    t = rc.TLAS()
    dv, dm = torch.from_numpy(verts).cuda(), torch.from_numpy(meta.view(np.int32)).cuda()
    assert t.add_geometry_device(dv.data_ptr(), len(verts), dm.data_ptr()) == 1
    t.push_instances(1)
    o = oracle.Scene()
    o.add_instance(o.add_blas(verts, meta))
    o.build()
    st = t.adapt()
    assert len(st.all_blas_prims) < len(verts)
    assert st.all_blas_prims.tobytes() == o.blas_prims.tobytes() and nodes_equal(st.all_blas_nodes, o.blas_nodes)


def test_build_parity_100k(rc, oracle):
    cfg = rc.scenes.config_c2()
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    assert nodes_equal(t.adapt().all_blas_nodes, o.blas_nodes)


def test_build_parity_1m_repeated(rc, oracle):
    """1 M triangles: the refit's cross-workgroup hand-off (agent-coherent stores, relaxed counter, coherent loads of the
    sibling box) runs under real contention across all 8 XCDs; every rebuild must still be byte-identical to the oracle."""
    import torch
    verts = rc.scenes.random_triangles(1_000_000, 4242, edge=0.01)
    o = oracle.Scene()
    o.add_instance(o.add_blas(verts))
    o.build()
    want = o.blas_nodes.tobytes()
    t = rc.TLAS()
    dv = torch.from_numpy(verts).cuda()
    for rep in range(3):
        b = t.add_geometry_device(dv.data_ptr(), len(verts))
        h = t.push_instances(b)
        st = t.adapt()
        got = st.all_blas_nodes
        # only the newest BLAS matters: it is the last n_nodes of the flat array
        assert got[-(2 * len(verts) - 1):].tobytes() == want, rep
        t.delete(h)   # the next sync compacts the old geometry away
    t.free()


# ---- traversal parity -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("kernel", [0, 1, 2, 3, 4, 5, 6, -1])
def test_trace_parity_c1(rc, oracle, kernel):
    cfg = rc.scenes.config_c1()
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    t.set_option("kernel", kernel)
    rays = o.ray_grid(cfg["viewdir"], cfg["grid"])
    got, want = t.trace(rays), o.trace(rays)
    assert 0 < want["hit"].sum() < len(rays)
    assert_hits_equal(got, want, "C1 closest")
    assert_hits_equal(t.trace(rays, mode="any"), o.trace(rays, mode="any"), "C1 any")


@pytest.mark.parametrize("kernel", [0, 1, 2, 3, 4, 5, 6, -1])
def test_trace_parity_random_scene(rc, oracle, kernel):
    sc = rc.scenes
    xf, _, _ = sc.lattice_transforms(3, 3, 2, 1.2, 77)
    cfg = {"blas": [(sc.random_triangles(5000, 5, lo=-0.5, hi=0.5, edge=0.08), None), (sc.fan_sphere(24, 13), None)],
           "instances": [(1, xf[:9], np.arange(9, dtype=np.uint32)), (2, xf[9:], np.arange(9, dtype=np.uint32) + 100)]}
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    t.set_option("kernel", kernel)
    wb = o.world_bound
    rays = random_rays(rc, 200_000, 11, wb[:3], wb[3:])
    rays["tmin"][::7] = 0.5           # t_min is honoured by closest_hit, ignored by any_hit (:1907 vs :2039)
    rays["tmax"][::5] = 2.0
    want = o.trace(rays, nthreads=8)
    assert 0.05 < want["hit"].mean() < 0.95
    assert_hits_equal(t.trace(rays), want, "random closest")
    assert_hits_equal(t.trace(rays, mode="any"), o.trace(rays, mode="any", nthreads=8), "random any")


@pytest.mark.parametrize("kernel", [1, 2, 3, 4, 5, 6, -1])
def test_trace_parity_c3_and_shadow(rc, oracle, kernel):
    cfg = rc.scenes.config_c3()
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    t.set_option("kernel", kernel)
    rays = rc.scenes.c3_primary_rays(cfg, 512, 512)
    got, want = t.trace(rays), o.trace(rays, nthreads=8)
    assert_hits_equal(got, want, "C3 primary")
    shadow = rc.scenes.c3_shadow_rays(cfg, rays, want)
    assert_hits_equal(t.trace(shadow, mode="any"), o.trace(shadow, mode="any", nthreads=8), "C3 shadow")
    bounce = rc.scenes.c4_bounce_rays(cfg, rays, want, 300_000)
    assert_hits_equal(t.trace(bounce), o.trace(bounce, nthreads=8), "C4 bounce")


@pytest.mark.parametrize("n_tris,n_inst", [(2, 1), (3, 4), (40, 1), (700, 1), (5000, 3), (5000, 200), (20000, 256)])
def test_blas_top_renumbering(rc, oracle, n_tris, n_inst):
    """Single-BLAS scenes: the traversal copy renumbers the BLAS's top internal nodes breadth-first (kernel 5 reads them from LDS).
    Every kernel walks that copy, so all of them must still agree with the oracle bit for bit; the exported arrays keep the
    reference numbering; a second BLAS switches the renumbering off."""
    sc = rc.scenes
    verts = sc.random_triangles(n_tris, 31 + n_tris, lo=-0.5, hi=0.5, edge=0.3 if n_tris < 100 else 0.08)
    xf, _, _ = sc.lattice_transforms(8, 8, 4, 1.1, 5)
    cfg = {"blas": [(verts, None)], "instances": [(1, xf[:n_inst], np.arange(n_inst, dtype=np.uint32))]}
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    n_int = len(o.blas_prims) - 1
    if os.environ.get("RC_STACK16", "1") != "0":   # (a campaign may run the suite with 32-bit lane stacks as the process default: the shape then has 310 plane entries)
        assert t.get_option("stack16_in_use") == 1   # every tree below 65 534 nodes: 16-bit lane stacks, 748 node-plane entries instead of 310
        assert t.get_option("blas_top_k") == min(n_int, 748 - (n_inst - 1))
    assert t.adapt().all_blas_nodes.tobytes() == o.blas_nodes.tobytes()
    wb = o.world_bound
    rays = random_rays(rc, 150_000, n_tris, wb[:3], wb[3:])
    want, want_any = o.trace(rays, nthreads=8), o.trace(rays, mode="any", nthreads=8)
    assert want["hit"].any()
    for kernel, top, s16 in ((5, 1, 1), (5, 0, 1), (0, 1, 1), (1, 1, 1), (3, 1, 1), (4, 1, 1), (6, 1, 1), (6, 0, 1), (5, 1, 0), (6, 1, 0), (-1, 1, 0)):
        t.set_option("kernel", kernel)
        t.set_option("blas_top", top)
        t.set_option("stack16", s16)     # 0: the 32-bit shape stages a PREFIX of the same renumbered top
        assert_hits_equal(t.trace(rays), want, f"kernel {kernel} blas_top {top} stack16 {s16} closest")
        assert_hits_equal(t.trace(rays, mode="any"), want_any, f"kernel {kernel} blas_top {top} stack16 {s16} any")
    t.set_option("blas_top", 1)
    t.set_option("stack16", 1)
    t.set_option("kernel", 5)
    h2 = t.push_instances(t.add_geometry(sc.fan_sphere(12, 7)))  # a second BLAS: plain copy again
    t.sync()
    assert t.get_option("blas_top_k") == 0
    o.add_instance(o.add_blas(sc.fan_sphere(12, 7)))
    o.build()
    assert_hits_equal(t.trace(rays), o.trace(rays, nthreads=8), "two BLASes")
    t.free()


@default_stack_shape
@pytest.mark.parametrize("n_tris,n_inst,n_blas", [(3000, 300, 1), (40, 700, 1), (5000, 1500, 1), (800, 600, 3), (2, 257, 1)])
def test_large_top_level_partial_lds(rc, oracle, n_tris, n_inst, n_blas):
    """More than 256 instances: the traversal copy's TLAS (and a single BLAS) is renumbered so that the breadth-first top sits in
    front and kernel 6 stages those nodes in LDS; transforms change through refits (the renumbering must survive them) and through a
    rebuild.  Every kernel must agree with the oracle on that copy; the exported arrays keep the reference numbering."""
    sc = rc.scenes
    g = np.random.default_rng(n_inst)
    blas = [(sc.random_triangles(n_tris, 7 + b, lo=-0.5, hi=0.5, edge=0.3 if n_tris < 100 else 0.1), None) for b in range(n_blas)]
    xf = np.tile(sc.IDENTITY3x4, (n_inst, 1)).astype(np.float32)
    xf[:, [3, 7, 11]] = g.uniform(-6, 6, size=(n_inst, 3))
    per = n_inst // n_blas
    inst = [(b + 1, xf[b * per:(b + 1) * per if b < n_blas - 1 else n_inst], np.arange(b * per, (b + 1) * per if b < n_blas - 1 else n_inst, dtype=np.uint32)) for b in range(n_blas)]
    cfg = {"blas": blas, "instances": inst}
    t = rc.TLAS(0)
    bids = [t.add_geometry(v, m) for v, m in cfg["blas"]]
    handles = [t.push_instances(b, x, i) for b, x, i in cfg["instances"]]
    t.sync()
    o = build_oracle(oracle, cfg)
    n_int_tlas = n_inst - 1
    n_int_blas = len(o.blas_prims) - 1 if n_blas == 1 else 0
    tk, bk = t.get_option("tlas_top_k"), t.get_option("blas_top_k")
    P = 1023 if t.get_option("stack16_in_use") else 585   # node-plane entries of kernel 6 (16-bit lane stacks leave room for 1023)
    assert P == 1023 and 0 < tk <= min(n_int_tlas, P) and bk <= min(n_int_blas, P) and tk + bk <= P
    assert tk + bk == min(P, n_int_tlas + n_int_blas)
    st = t.adapt()
    assert st.nodes.tobytes() == o.tlas_nodes.tobytes() and st.all_blas_nodes.tobytes() == o.blas_nodes.tobytes()
    wb = o.world_bound
    rays = random_rays(rc, 120_000, n_inst, wb[:3], wb[3:])

    def check_all(what):
        want, want_any = o.trace(rays, nthreads=8), o.trace(rays, mode="any", nthreads=8)
        assert want["hit"].any()
        for kernel, s16 in ((6, 1), (6, 0), (3, 1), (0, 1), (1, 1), (5, 1)):  # 5 falls back to 3 here (too many instances); stack16 0: kernel 6 stages a 585-entry prefix of the renumbered tops
            t.set_option("kernel", kernel); t.set_option("stack16", s16)
            assert_hits_equal(t.trace(rays), want, f"{what} kernel {kernel} stack16 {s16} closest")
            assert_hits_equal(t.trace(rays, mode="any"), want_any, f"{what} kernel {kernel} stack16 {s16} any")
        t.set_option("kernel", -1); t.set_option("stack16", 1)

    check_all("built")
    xf2 = xf.copy()
    xf2[:, [3, 7, 11]] += g.uniform(-1, 1, size=(n_inst, 3)).astype(np.float32)
    for (b, x, i), h in zip(cfg["instances"], handles):
        lo = int(i[0])
        t.update_transforms(h, xf2[lo:lo + len(i)])
    t.sync()
    assert t.last_sync_action == "refit" and t.get_option("tlas_top_k") == tk
    o2 = oracle.Scene()
    for v, m in cfg["blas"]:
        o2.add_blas(v, m)
    for (b, x, i) in cfg["instances"]:
        for k, idx in enumerate(i):
            o2.add_instance(b, xf2[int(idx)], int(idx))
    o2.build()
    # a refit keeps the topology of the first build: compare against the oracle only through the rays (the refitted tree is a valid
    # BVH of the moved instances, and every kernel walks the same one)
    t.set_option("kernel", 3)
    ref = t.trace(rays)
    for kernel in (6, 0, 1):
        t.set_option("kernel", kernel)
        assert_hits_equal(t.trace(rays), ref, f"refit kernel {kernel}")
    t.set_option("kernel", -1)
    brute = o2.trace(rays, nthreads=8)
    assert np.array_equal(ref["hit"], brute["hit"]) and np.array_equal(ref["t"][ref["hit"] == 1], brute["t"][brute["hit"] == 1])
    t.free()


def test_trace_edge_cases(rc, oracle):
    """Axis-parallel rays, -0 directions, rays in the plane of a triangle (det == 0 => NaN-t 'hit',
    SURVEY.md Appendix A), duplicate instances (exact t ties: the later visit wins), t_max culling."""
    quad = np.array([[0, 0, 0, 1, 0, 0, 1, 1, 0], [0, 0, 0, 1, 1, 0, 0, 1, 0]], dtype=np.float32)
    wall = np.array([[0.5, 0, -1, 0.5, 1, -1, 0.5, 0, 1], [0.5, 1, -1, 0.5, 1, 1, 0.5, 0, 1]], dtype=np.float32)
    cfg = {"blas": [(quad, [1, 2]), (wall, [3, 4])],
           "instances": [(1, np.stack([rc.scenes.IDENTITY3x4] * 5), np.arange(5, dtype=np.uint32)),
                         (2, rc.scenes.IDENTITY3x4[None], np.array([9], np.uint32))]}
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    g = rc.scenes.rng(3)
    os_, ds = [], []
    for x in np.linspace(-0.25, 1.25, 31):
        for y in np.linspace(-0.25, 1.25, 31):
            os_.append([x, y, 1.0]); ds.append([0.0, -0.0, -1.0])     # straight down, -0 component
            os_.append([x, y, -2.0]); ds.append([0.0, 0.0, 1.0])
            os_.append([0.5, y, 3.0]); ds.append([0.0, 0.0, -1.0])     # in the plane of the wall: det == 0
            os_.append([x, 0.5, 0.0]); ds.append([1.0, 0.0, 0.0])      # in the plane of the quad
    rays = rc.scenes.make_rays(os_, ds)
    want = o.trace(rays)
    for k in (-1, 0, 1, 2, 3, 4, 5, 6):
        t.set_option("kernel", k)
        assert_hits_equal(t.trace(rays), want, f"edge closest k{k}")
        assert_hits_equal(t.trace(rays, mode="any"), o.trace(rays, mode="any"), f"edge any k{k}")
    t.set_option("kernel", -1)
    hit = want["hit"] == 1
    assert np.isnan(want["t"][hit]).any()            # the in-plane rays: det == 0 => u = NaN passes every test => a "hit" with t = NaN
    quad_hits = hit & ~np.isnan(want["t"]) & (want["instance_id"] < 5)
    assert quad_hits.any() and len(np.unique(want["instance_id"][quad_hits])) < 5  # five identical instances: every t ties and ONE visit order decides (the later visit replaces, :1792)


def test_weird_rays_and_scales(rc, oracle):
    """Inputs outside the comfortable range, still required to match the oracle bit for bit: zero / infinite / tiny
    direction components (safe_invdir clamps, :1742-1748), t_max < t_min, negative t_min, huge and tiny coordinates,
    NaN-free but extreme rays (NaN / Inf rays: test_nan_and_inf_rays)."""
    sc = rc.scenes
    xf, _, _ = sc.lattice_transforms(2, 2, 2, 2.0, 5)
    xf[3, [0, 5, 10]] *= 1e3      # a huge instance
    xf[4, [0, 5, 10]] *= 1e-3     # and a tiny one
    xf[5, [3, 7, 11]] += 1e4      # far from the origin
    cfg = {"blas": [(sc.fan_sphere(12, 7, radius=0.5), None), (sc.random_triangles(100, 2, lo=-0.5, hi=0.5, edge=0.4), None)],
           "instances": [(1, xf[:5], np.arange(5, dtype=np.uint32)), (2, xf[5:], np.arange(3, dtype=np.uint32) + 10)]}
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    g = sc.rng(77)
    n = 20000
    org = g.uniform(-3, 5, size=(n, 3))
    d = sc.normalize(g.normal(size=(n, 3)))
    rays = sc.make_rays(org, d)
    k = n // 10
    rays["d"][0 * k:1 * k, 0] = 0.0                       # exactly axis-plane directions
    rays["d"][1 * k:2 * k, 1] = -0.0
    rays["d"][2 * k:3 * k] *= np.float32(1e-7)            # |d| tiny: every component is clamped to +-1e-5
    rays["d"][3 * k:4 * k, 2] = np.float32(1e-6)          # one clamped component
    rays["d"][4 * k:5 * k] *= np.float32(1e6)             # long direction vectors (t is in units of |d|)
    rays["tmin"][5 * k:6 * k] = 2.0
    rays["tmax"][5 * k:6 * k] = 1.0                       # empty interval
    rays["tmin"][6 * k:7 * k] = -5.0                      # hits behind the origin are legal for closest_hit
    rays["o"][7 * k:8 * k] = (xf[5, [3, 7, 11]] + g.uniform(-2, 2, size=(k, 3))).astype(np.float32)
    rays["o"][8 * k:9 * k] *= np.float32(1e3)
    rays["d"][9 * k:9 * k + 50] = [0.0, 0.0, 0.0]         # null direction
    rays["d"][9 * k + 50:9 * k + 100] = [np.inf, 0.0, 0.0]
    want_c, want_a = o.trace(rays, nthreads=8), o.trace(rays, mode="any", nthreads=8)
    assert 0 < want_c["hit"].sum() < n
    for kern in (-1, 0, 1, 2, 3, 4, 5, 6):
        t.set_option("kernel", kern)
        got_c, got_a = t.trace(rays), t.trace(rays, mode="any")
        ok = ~(np.isnan(want_c["t"]) | np.isnan(got_c["t"]))  # NaN != NaN bitwise is fine to compare too, but keep ids strict
        assert_hits_equal(got_c, want_c, f"weird closest k{kern}")
        assert_hits_equal(got_a, want_a, f"weird any k{kern}")
        assert ok.any()


def test_nan_and_inf_rays(rc, oracle):
    """Rays with NaN / Inf components.  Under Julia's NaN-propagating min/max a NaN slab component fails every box test of
    that level, while a triangle reached without a box test (single-leaf TLAS / BLAS) is 'hit' with NaN t (all comparisons
    false).  The kernels reproduce both: their slab test runs on v_minimum3_f32 / v_maximum3_f32, which propagate
    NaNs exactly like Julia's min / max (rc_device.h: jl_minf / jl_maxf)."""
    sc = rc.scenes
    xf, _, _ = sc.lattice_transforms(2, 2, 1, 2.0, 9)
    multi = {"blas": [(sc.fan_sphere(10, 6, radius=0.6), None)], "instances": [(1, xf, np.arange(4, dtype=np.uint32))]}
    single = {"blas": [(UNIT_TRI, [7])], "instances": [(1, sc.IDENTITY3x4[None], np.array([3], np.uint32))]}  # TLAS root and BLAS root are leaves
    rot = sc.IDENTITY3x4.copy()
    rot[[0, 1, 4, 5]] = [0.0, -1.0, 1.0, 0.0]  # 90 degrees about z: zeros in the matrix meet Inf components => NaN after the transform
    two = {"blas": [(UNIT_TRI, [1])], "instances": [(1, np.stack([sc.IDENTITY3x4, rot]), np.array([0, 1], np.uint32))]}
    g = sc.rng(5)
    n = 6000
    rays = sc.make_rays(g.uniform(-1, 3, size=(n, 3)), sc.normalize(g.normal(size=(n, 3))))
    rays["o"][:n // 2] = [0.25, 0.25, 1.0]
    rays["d"][:n // 2] = [0.0, 0.0, -1.0]
    bad = [np.nan, np.inf, -np.inf]
    for i in range(0, n, 3):
        f = ("o", "d")[(i // 3) % 2]
        rays[f][i, (i // 6) % 3] = bad[(i // 18) % 3]
    rays["tmin"][1::12] = np.nan
    rays["tmax"][2::12] = np.nan
    rays["tmax"][5::12] = np.inf
    rays["tmin"][8::12] = -np.inf
    # NaN / Inf inside the geometry: the NaN climbs into every ancestor box (Julia's min / max propagate it) and those boxes fail
    # every test; the kernels' v_minimum3 / v_maximum3 slab test does the same
    bad_v = sc.random_triangles(400, 6, lo=-1, hi=1, edge=0.4)
    bad_v[7, 0] = np.nan
    bad_v[100, 5] = np.inf
    bad_v[200, 7] = -np.inf
    nan_geom = {"blas": [(bad_v, None)], "instances": [(1, xf[:2], np.arange(2, dtype=np.uint32))]}
    for name, cfg in (("multi", multi), ("single", single), ("two", two), ("nan_geometry", nan_geom)):
        t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
        want_c, want_a = o.trace(rays, nthreads=4), o.trace(rays, mode="any", nthreads=4)
        for kern in (-1, 0, 1, 2, 3, 4, 5, 6):
            t.set_option("kernel", kern)
            assert_hits_equal(t.trace(rays), want_c, f"nan {name} closest k{kern}")
            assert_hits_equal(t.trace(rays, mode="any"), want_a, f"nan {name} any k{kern}")
        if name == "single":
            assert np.isnan(want_c["t"][want_c["hit"] == 1]).any()  # NaN-t hits exist and are reproduced bit for bit


@default_stack_shape
def test_deep_trees_use_the_stack_spill_path(rc, oracle):
    """LBVH chains (one leaf split off per level: Morton codes that are successive powers of two) make 30-level BLAS and
    deep TLAS trees; rays through the shared corner keep one pending far child per level, so the per-lane stack outgrows
    its 24 LDS entries and runs through the global spill area (the reference's 32-entry MVector would overflow, :1912)."""
    def chain(levels, fat=0.3):
        tris = []
        for j in range(1, levels + 1):
            for axis in range(3):
                size = 2.0 ** (-j + 1)
                p = np.full(3, fat * size); p[axis] = size
                q = p.copy(); q[(axis + 1) % 3] += 0.5 * size * fat
                r = np.full(3, -1e-4 * (1 + 0.5 * j))   # smaller triangles start earlier along +rays: the deep subtree is the near child
                tris.append(np.concatenate([r, p, q]))
        return np.array(tris, dtype=np.float32)
    verts = chain(10)
    xf = np.tile(rc.scenes.IDENTITY3x4, (4, 1)).astype(np.float32)
    xf[1, [0, 5, 10]] = 0.5
    xf[2, [0, 5, 10]] = 0.25
    xf[3, [3, 7, 11]] = [0.01, 0.0, 0.0]
    cfg = {"blas": [(verts, None)], "instances": [(1, xf, np.arange(4, dtype=np.uint32))]}
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    g = rc.scenes.rng(12)
    n = 20000
    org = g.uniform(-0.2, 0.0, size=(n, 3))
    d = rc.scenes.normalize(g.uniform(0.05, 1.0, size=(n, 3)))
    rays = rc.scenes.make_rays(org, d)
    want = o.trace(rays, nthreads=8)
    assert want["hit"].mean() > 0.01
    t.set_option("kernel", 1); t.set_option("stats", 1)
    assert_hits_equal(t.trace(rays), want, "deep k1")
    max_sp = t.get_option("stat2")
    t.set_option("stats", 0)
    assert max_sp > 24, f"scene too shallow to reach the spill path (max stack {max_sp})"
    assert t.get_option("stack16_in_use") == 1   # small trees: kernels 5 / 6 keep 16-bit lane-stack entries in LDS; the spill area beyond the LDS depth holds the same values in 32-bit words
    want_any = o.trace(rays, mode="any", nthreads=8)
    for k, s16 in ((-1, 1), (0, 1), (2, 1), (3, 1), (4, 1), (5, 1), (6, 1), (5, 0), (6, 0)):
        t.set_option("kernel", k); t.set_option("stack16", s16)
        assert_hits_equal(t.trace(rays), want, f"deep k{k} stack16 {s16}")
        assert_hits_equal(t.trace(rays, mode="any"), want_any, f"deep any k{k} stack16 {s16}")


def test_full_size_c2_properties(rc, oracle):
    """BASELINE C2 at full size: 100k triangles, 1M coherent rays.  Oracle comparison on every ray (the C
    oracle does 1M rays in seconds on 8 threads) plus size-independent properties."""
    cfg = rc.scenes.config_c2()
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    rays = o.ray_grid(cfg["viewdir"], cfg["grid"])
    got = t.trace(rays)
    assert_hits_equal(got, o.trace(rays, nthreads=8), "C2 full")
    # any_hit == "closest_hit with t_min = 0 hits" (any_hit ignores t_min)
    anyh = t.trace(rays, mode="any")
    assert np.array_equal(anyh["hit"], got["hit"])
    # permutation invariance: per-ray results do not depend on scheduling
    perm = rc.scenes.rng(5).permutation(len(rays))
    assert_hits_equal(t.trace(rays[perm]), got[perm], "C2 permuted")
    # every kernel variant agrees
    for k in (0, 1, 2, 3, 4, 5, 6):
        t.set_option("kernel", k)
        assert_hits_equal(t.trace(rays), got, f"C2 kernel {k}")
        assert np.array_equal(t.trace(rays, mode="any")["hit"], got["hit"])


def test_full_size_c3_c4_properties(rc, oracle):
    """BASELINE C3 (4 M primary rays + shadow rays) and C4 (16 M incoherent bounce rays) at full size on the device:
    EVERY ray against the oracle (one oracle thread per host CPU: 23 M rays are seconds on the GPU box's cores), and
    size-independent properties over the whole batch -- every kernel variant agrees bit for bit, permuting the batch
    permutes the results, any_hit agrees with closest_hit on occlusion."""
    import os
    import torch
    cpus = os.cpu_count() or 8
    cfg = rc.scenes.config_c3()
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    rays = rc.scenes.c3_primary_rays(cfg, 2048, 2048)
    n = len(rays)

    def run(rs, mode, kernel):
        t.set_option("kernel", kernel)
        dr = torch.from_numpy(rs.view(np.uint8).reshape(-1)).cuda()
        dh = torch.empty(len(rs) * 32, dtype=torch.uint8, device="cuda")
        t.trace_device(dr.data_ptr(), dh.data_ptr(), len(rs), mode=mode)
        torch.cuda.synchronize()
        return dh.cpu().numpy().view(rc.HIT_DT)

    prim = run(rays, "closest", -1)
    assert_hits_equal(prim, o.trace(rays, nthreads=cpus), "C3 primary, all 4 194 304 rays")
    assert_hits_equal(run(rays, "closest", 0), prim, "C3 kernel 0 vs default")
    shadow = rc.scenes.c3_shadow_rays(cfg, rays, prim)
    occ = run(shadow, "any", -1)
    assert_hits_equal(occ, o.trace(shadow, mode="any", nthreads=cpus), "C3 shadow rays, all of them")
    assert np.array_equal(occ["hit"], run(shadow, "closest", -1)["hit"])   # occluded <=> a closest hit exists within t_max
    bounce = rc.scenes.c4_bounce_rays(cfg, rays, prim, 4 * n)
    assert len(bounce) == 16_777_216
    b3 = run(bounce, "closest", 3)
    assert_hits_equal(b3, o.trace(bounce, nthreads=cpus), "C4 bounce rays, all 16 777 216")
    assert_hits_equal(run(bounce, "closest", -1), b3, "C4 default kernel vs 3")
    assert_hits_equal(run(bounce, "closest", 1), b3, "C4 kernel 1 vs 3")
    perm = rc.scenes.rng(3).permutation(len(bounce))
    assert_hits_equal(run(bounce[perm], "closest", 3), b3[perm], "C4 permuted")
    t.set_option("kernel", -1)


# ---- drivers ---------------------------------------------------------------------------------------------------
def test_ray_grid_and_illumination_parity(rc, oracle):
    for cfg, grid in ((rc.scenes.config_c1(), 64), (rc.scenes.config_c2(20_000, 300), 300)):
        t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
        rays = rc.generate_ray_grid(t, cfg["viewdir"], grid)
        assert rays.tobytes() == o.ray_grid(cfg["viewdir"], grid).tobytes()
        got, want = rc.get_illumination(t, cfg["viewdir"], grid), o.get_illumination(cfg["viewdir"], grid, nthreads=8)
        assert got.dtype == np.float32 and np.array_equal(got, want)
        t.set_option("kernel", 3)  # the 256-thread driver kernel without LDS node planes (what large top levels get)
        assert np.array_equal(rc.get_illumination(t, cfg["viewdir"], grid), want)
        t.set_option("kernel", -1)
        h = t.trace(rays)
        metas = t.adapt().all_blas_prims["meta"][h["primitive_id"][h["hit"] == 1]]
        assert got.sum() == np.count_nonzero(metas <= len(got))  # metadata outside 1..N is dropped (src/kernels.jl:123)


def test_drivers_on_a_large_top_level(rc, oracle):
    """More than 256 instances: get_illumination and view_factors run the partial-LDS driver kernels (tops of the TLAS and of the single
    BLAS staged, the shape of trace kernel 6); same counts as the oracle and as the plain 256-thread kernels."""
    sc = rc.scenes
    g = np.random.default_rng(5)
    n_inst = 400
    xf = np.tile(sc.IDENTITY3x4, (n_inst, 1)).astype(np.float32)
    xf[:, [3, 7, 11]] = g.uniform(-6, 6, size=(n_inst, 3)).astype(np.float32)
    verts = sc.fan_sphere(6, 4, radius=0.6)  # 36 triangles per instance
    n = len(verts)
    cfg = {"blas": [(verts, np.arange(1, n + 1, dtype=np.uint32))], "instances": [(1, xf.reshape(n_inst, 3, 4), np.arange(n_inst, dtype=np.uint32))]}
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    assert t.get_option("tlas_top_k") > 0 and t.get_option("blas_top_k") > 0
    want_i = o.get_illumination((0.2, -0.1, -1.0), 300, nthreads=8)
    want_v = o.view_factors(64, seed=99, nthreads=8)
    for kernel in (-1, 3):
        t.set_option("kernel", kernel)
        assert np.array_equal(rc.get_illumination(t, (0.2, -0.1, -1.0), 300), want_i), kernel
        assert np.array_equal(rc.view_factors(t, rays_per_triangle=64, seed=99), want_v), kernel
    assert want_i.sum() > 1000 and want_v.sum() > 100
    t.free()


def test_illumination_hot_counters(rc, oracle):
    """Most rays land on two large triangles: the histogram's wave-level aggregation (equal targets among the lanes that finish
    together become one atomic) must still give the reference's counts exactly."""
    sc = rc.scenes
    ground = np.array([[-50, -50, 0, 50, -50, 0, 50, 50, 0], [-50, -50, 0, 50, 50, 0, -50, 50, 0]], dtype=np.float32)
    verts = np.concatenate([ground, sc.fan_sphere(24, 13, centre=(0, 0, 3), radius=2.0)]).astype(np.float32)
    cfg = {"blas": [(verts, np.arange(1, len(verts) + 1, dtype=np.uint32))], "instances": [(1, sc.IDENTITY3x4[None], np.zeros(1, np.uint32))]}
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    got, want = rc.get_illumination(t, [0.05, 0.02, -1.0], 400), o.get_illumination([0.05, 0.02, -1.0], 400, nthreads=8)
    assert np.array_equal(got, want) and got[:2].sum() > 0.5 * got.sum() > 0
    t.set_option("kernel", 3)
    assert np.array_equal(rc.get_illumination(t, [0.05, 0.02, -1.0], 400), want)
    t.free()


def test_hits_from_grid_and_centroid(rc, oracle):  # src/kernels.jl:58-72, 106-110
    cfg = rc.scenes.config_c1()
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    g = 48
    hg = rc.hits_from_grid(t, cfg["viewdir"], g)
    assert hg.shape == (g, g)
    rays = o.ray_grid(cfg["viewdir"], g)
    want = o.trace(rays)
    flat = hg.reshape(-1, order="F")
    assert np.array_equal(flat["hit"], want["hit"] == 1)
    m = want["hit"] == 1
    prims = o.blas_prims[want["primitive_id"][m]]
    u, v = want["bary_u"][m], want["bary_v"][m]
    w = (np.float32(1) - u) - v
    pts = (w[:, None] * prims["v"][:, 0] + u[:, None] * prims["v"][:, 1]) + v[:, None] * prims["v"][:, 2]
    assert flat["point"][m].tobytes() == pts.astype(np.float32).tobytes()
    assert np.array_equal(flat["metadata"][m], prims["meta"])
    # the sphere is centred at (0,0,2) with radius 1: hit points lie on it, their mean is on the axis towards the viewer
    assert np.allclose(np.linalg.norm(flat["point"][m] - [0, 0, 2], axis=1), 1.0, atol=0.02)
    pts2, mean = rc.get_centroid(t, cfg["viewdir"], g)
    assert len(pts2) == m.sum() and abs(mean[0]) < 0.02 and abs(mean[1]) < 0.02 and mean[2] < 2.0


def test_view_factor_rays_bit_exact(rc, oracle):
    """The sampled rays themselves (Philox -> triangle point -> hemisphere direction, src/kernels.jl:83-92,
    src/math.jl:125-174) are bit-identical on device and oracle, not just the counted matrices."""
    import torch
    from raycore_jl_amd._capi import check, lib, ptr
    sc = rc.scenes
    verts = np.concatenate([sc.fan_sphere(12, 7, centre=(0.3, -0.2, 0.1), radius=0.5), sc.random_triangles(300, 4, lo=-1, hi=1, edge=0.3)])
    cfg = {"blas": [(verts, np.arange(1, len(verts) + 1, dtype=np.uint32))], "instances": [(1, sc.IDENTITY3x4[None], np.zeros(1, np.uint32))]}
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    n_ray = 512
    buf = torch.empty(n_ray * 32, dtype=torch.uint8, device="cuda")
    for src in (0, 1, 17, 100, len(verts) - 1):
        check(lib().rc_view_factor_rays_device(t._h, 99, src, 5, n_ray, ptr(buf.data_ptr()), None))
        torch.cuda.synchronize()
        got = buf.cpu().numpy().view(rc.RAY_DT)
        want = np.array([o.view_factor_ray(src, 5 + i, seed=99) for i in range(n_ray)], dtype=rc.RAY_DT)
        assert got.tobytes() == want.tobytes(), src


def test_view_factors_parity(rc, oracle):
    sc = rc.scenes
    verts = np.concatenate([sc.fan_sphere(12, 7, centre=(0, 0, 0), radius=0.5), sc.box_room((-1.5, -1.5, -1.5), (1.5, 1.5, 1.5), 2)])
    n = len(verts)
    cfg = {"blas": [(verts, np.arange(1, n + 1, dtype=np.uint32))], "instances": [(1, sc.IDENTITY3x4[None], np.zeros(1, np.uint32))]}
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    got = rc.view_factors(t, rays_per_triangle=256, seed=1234)
    want = o.view_factors(256, seed=1234, nthreads=8)
    assert got.shape == (n, n) and got.dtype == np.uint32
    assert np.array_equal(got, want)
    t.set_option("kernel", 3)  # the 256-thread driver kernel without LDS node planes
    assert np.array_equal(rc.view_factors(t, rays_per_triangle=256, seed=1234), want)
    t.set_option("kernel", -1)
    assert got.sum() > 0 and np.all(np.diag(got) == 0)
    assert np.all(got.sum(axis=1) <= 256)
    # sharding independence: two half-jobs accumulate to the whole (Philox keyed by (src, ray))
    import torch
    m = torch.zeros(n * n, dtype=torch.int32, device="cuda")
    from raycore_jl_amd._capi import check, lib, ptr
    check(lib().rc_view_factors_device(t._h, 256, 1234, 0, n // 2, 0, 256, ptr(m.data_ptr()), 1, n, 0, 0, None))
    check(lib().rc_view_factors_device(t._h, 256, 1234, n // 2, n, 0, 100, ptr(m.data_ptr()), 1, n, 0, 0, None))
    check(lib().rc_view_factors_device(t._h, 256, 1234, n // 2, n, 100, 256, ptr(m.data_ptr()), 1, n, 0, 0, None))
    torch.cuda.synchronize()
    assert np.array_equal(m.cpu().numpy().view(np.uint32).reshape(n, n).T, got)  # column-major [src + N*dst]
    # the multi-GPU driver on one rank: both partitions reproduce the matrix (row-major [src][dst])
    from raycore_jl_amd import distributed as rd
    for mode in ("rays", "rows"):
        out = rd.view_factors_distributed(t, 256, 1234, mode=mode)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint32), got), mode
    block, rows = rd.view_factors_distributed(t, 256, 1234, mode="rows_sharded")
    torch.cuda.synchronize()
    full = np.zeros_like(got)
    full[rows] = block.cpu().numpy().view(np.uint32)
    assert np.array_equal(full, got)
    il = rd.get_illumination_distributed(t, [0.3, 0.2, 1.0], 128)
    torch.cuda.synchronize()
    assert np.array_equal(il.cpu().numpy(), rc.get_illumination(t, [0.3, 0.2, 1.0], 128))


def test_wavefront_stages(rc, oracle):
    """primary trace -> hit points / normals -> shadow rays -> any_hit, all on device buffers, vs the oracle's stages."""
    import torch
    cfg = rc.scenes.config_c3()
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    rays = rc.scenes.c3_primary_rays(cfg, 256, 256)
    n = len(rays)
    d_rays = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
    d_hits = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
    d_shadow = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
    d_occ = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
    d_pts, d_nrm = torch.empty(n * 3, dtype=torch.float32, device="cuda"), torch.empty(n * 3, dtype=torch.float32, device="cuda")
    light = cfg["light"].astype(np.float32)
    t.trace_device(d_rays.data_ptr(), d_hits.data_ptr(), n)
    t.hit_points_device(d_rays.data_ptr(), d_hits.data_ptr(), n, d_pts.data_ptr(), d_nrm.data_ptr())
    t.shadow_rays_device(d_rays.data_ptr(), d_hits.data_ptr(), n, light, d_shadow.data_ptr(), bias=0.01)
    t.trace_device(d_shadow.data_ptr(), d_occ.data_ptr(), n, mode="any")
    torch.cuda.synchronize()
    hits = o.trace(rays, nthreads=8)
    assert_hits_equal(d_hits.cpu().numpy().view(rc.HIT_DT), hits, "stage primary")
    pts, nrm = o.hit_points(rays, hits)
    assert d_pts.cpu().numpy().reshape(n, 3).tobytes() == pts.tobytes()
    assert d_nrm.cpu().numpy().reshape(n, 3).tobytes() == nrm.tobytes()
    shadow = o.shadow_rays(rays, hits, light, 0.01)
    assert d_shadow.cpu().numpy().view(rc.RAY_DT).tobytes() == shadow.tobytes()
    occ = o.trace(shadow, mode="any", nthreads=8)
    assert_hits_equal(d_occ.cpu().numpy().view(rc.HIT_DT), occ, "stage shadow any_hit")
    lit = (hits["hit"] == 1) & (occ["hit"] == 0)
    assert 0 < lit.sum() < (hits["hit"] == 1).sum()


# ---- lifecycle (handles, dirty flags, refit identity, errors) -------------------------------------------------
def test_primary_rays_and_hit_compaction(rc, oracle):
    """generate_primary_rays_lookat! (docs/src/wavefront-renderer.jl:219-254) bit-exact against the oracle, and the hit-index
    compaction stage against numpy."""
    import torch
    cfg = rc.scenes.config_c1()
    t = build_product(rc, cfg)
    pos, fwd = np.array([0.1, -0.2, -3.0], np.float32), np.array([0.0, 0.05, 1.0], np.float32)
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, [0, 1, 0]).astype(np.float32)
    right /= np.linalg.norm(right)
    up = np.cross(right, fwd).astype(np.float32)
    w, h, spp = 97, 61, 3
    for jitter in (True, False):
        d_r = torch.zeros(w * h * spp * 32, dtype=torch.uint8, device="cuda")
        t.primary_rays_lookat_device(pos, right, up, fwd, 0.4, 0.25, w, h, d_r.data_ptr(), samples=spp, seed=0xABCDEF0123, jitter=jitter)
        torch.cuda.synchronize()
        got = d_r.cpu().numpy().view(rc.RAY_DT)
        want = oracle.primary_rays_lookat(pos, right, up, fwd, 0.4, 0.25, w, h, spp, 0xABCDEF0123, jitter)
        assert got.tobytes() == want.tobytes()
    assert np.all(np.abs(np.linalg.norm(got["d"], axis=1) - 1) < 1e-6)
    hits = t.trace(got)
    assert 0 < hits["hit"].sum() < len(hits)
    d_h = torch.from_numpy(hits.view(np.uint8).reshape(-1)).cuda()
    d_idx = torch.full((len(hits),), -1, dtype=torch.int32, device="cuda")
    d_cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    t.compact_hits_device(d_h.data_ptr(), len(hits), d_idx.data_ptr(), d_cnt.data_ptr())
    torch.cuda.synchronize()
    want_idx = np.nonzero(hits["hit"])[0]
    assert int(d_cnt.item()) == len(want_idx) and np.array_equal(d_idx.cpu().numpy()[:len(want_idx)], want_idx)
    t.compact_hits_device(d_h.data_ptr(), 0, d_idx.data_ptr(), d_cnt.data_ptr())
    torch.cuda.synchronize()
    assert int(d_cnt.item()) == 0


def test_lifecycle_handles_and_errors(rc):  # test/test_instanced_bvh.jl:417-589, test/test_tlas_stress.jl:585-617
    m1, m2 = UNIT_TRI, UNIT_TRI + np.tile([5, 0, 0], 3).astype(np.float32)
    t, hs = rc.TLAS_from_meshes([m1, m2])
    assert len(hs) == 2 and t.n_instances(hs[0]) == 1 and t.is_valid(hs[0]) and t.n_geometries() == 2 and t.n_instances() == 2
    h3 = t.push(m1, [xlat(0, 0, 0), xlat(2, 0, 0)])
    assert t.n_instances(h3) == 2
    t.sync()
    assert t.last_sync_action == "rebuild" and t.n_geometries() == 3 and t.n_instances() == 4
    assert t.sync().last_sync_action == "noop"
    t.update_transforms(h3, [xlat(1, 0, 0), xlat(3, 0, 0)])
    assert np.allclose(t.get_instances(h3)["transform"][:, 3], [1, 3])
    assert t.sync().last_sync_action == "refit"
    with pytest.raises(rc.RaycoreError):
        t.update_transform(h3, xlat(0, 0, 0))      # 2 instances: must use update_transforms
    with pytest.raises(rc.RaycoreError):
        t.update_transforms(h3, [xlat(0, 0, 0)])   # arity mismatch
    assert t.delete(hs[0]) and not t.is_valid(hs[0]) and t.n_instances() == 3 and t.n_total_instances() == 4
    with pytest.raises(rc.RaycoreError):
        t.update_transform(hs[0], xlat(0, 0, 0))   # deleted handle
    with pytest.raises(rc.RaycoreError):
        t.get_instance(rc.TLASHandle(999))
    with pytest.raises(rc.RaycoreError):
        t.trace(rc.scenes.make_rays([[0, 0, 1]], [0, 0, -1]))  # pending mutation: must sync (adapt) first
    t.sync()
    assert t.n_instances() == 3 and t.n_total_instances() == 3 and t.n_geometries() == 2 and t.is_valid(hs[1])
    with pytest.raises(rc.RaycoreError):
        rc.TLAS().push(np.array([[0, 0, 0, 1, 0, 0, 2, 0, 0]], np.float32))  # only degenerate faces


def test_lifecycle_matches_fresh_oracle_scene(rc, oracle):
    """After push / delete / update_transform / update(geometry) + sync the traced results equal those of an
    oracle scene built from scratch with the surviving instances (handle-id order)."""
    sc = rc.scenes
    sphere, blob = sc.fan_sphere(16, 9), sc.random_triangles(800, 9, lo=-0.5, hi=0.5, edge=0.1)
    xf, _, _ = sc.lattice_transforms(4, 2, 1, 1.5, 21)
    t = rc.TLAS()
    h1 = t.push(sphere, xf[:3], instance_ids=[1, 2, 3])
    h2 = t.push(blob, xf[3:5], instance_ids=[4, 5])
    h3 = t.push(sphere, xf[5:8], instance_ids=[6, 7, 8])
    t.sync()
    t.delete(h2)
    t.sync()
    xf2, _, _ = sc.lattice_transforms(3, 1, 1, 1.7, 22)
    t.update_transforms(h3, xf2)
    t.sync()
    assert t.last_sync_action == "refit"
    t.update(h1, blob)
    t.sync()
    o = oracle.Scene()
    b1, b3 = o.add_blas(blob), o.add_blas(sphere)
    for x, i in zip(xf[:3], (1, 2, 3)):
        o.add_instance(b1, x, i)
    for x, i in zip(xf2, (6, 7, 8)):
        o.add_instance(b3, x, i)
    o.build()
    st = t.adapt()
    assert st.instances.tobytes() == o.instances.tobytes()
    assert nodes_equal(st.nodes, o.tlas_nodes) and nodes_equal(st.all_blas_nodes, o.blas_nodes)
    wb = o.world_bound
    rays = random_rays(rc, 50_000, 4, wb[:3], wb[3:])
    assert_hits_equal(t.trace(rays), o.trace(rays, nthreads=8), "lifecycle")


def test_refit_equals_rebuild_boxes(rc, oracle):  # refit_tlas! keeps topology, refreshes boxes (src/instanced-bvh.jl:2197-2222)
    sc = rc.scenes
    xf, _, _ = sc.lattice_transforms(5, 5, 2, 1.5, 31)
    t = rc.TLAS()
    h = t.push(sc.fan_sphere(12, 7), xf)
    t.sync()
    topo_before = t.adapt().nodes[["child0", "child1", "parent"]].copy()
    moved = xf.copy()
    moved[:, 3] += 0.05
    t.update_transforms(h, moved)
    t.sync()
    assert t.last_sync_action == "refit"
    n = t.adapt().nodes
    assert np.array_equal(n[["child0", "child1", "parent"]], topo_before)
    rays = random_rays(rc, 20_000, 6, t.world_bound().p_min, t.world_bound().p_max)
    o = oracle.Scene()
    b = o.add_blas(sc.fan_sphere(12, 7))
    for x in moved:
        o.add_instance(b, x, 0)
    o.build()
    assert np.array_equal(np.concatenate(t.world_bound()), o.world_bound)
    assert_hits_equal(t.trace(rays), o.trace(rays, nthreads=8), "refit")


@default_stack_shape
def test_stack16_applies_only_where_every_node_index_fits(rc, oracle):
    """Round 5: scenes whose trees ALL have fewer than 65 534 nodes run kernels 5 / 6 with 16-bit lane-stack entries (INVALID and the sentinel
    are their own low halves there); one BLAS of 32 768 triangles, or 32 768 instances, and the scene keeps the 32-bit shape.  Results are
    the oracle's on both sides of the limit and with the option off."""
    sc = rc.scenes
    for n_tris, expect in ((32767, 1), (32768, 0)):
        verts = sc.random_triangles(n_tris, 77, lo=-0.5, hi=0.5, edge=0.05)
        xf, _, _ = sc.lattice_transforms(2, 2, 1, 1.2, 9)
        cfg = {"blas": [(verts, None)], "instances": [(1, xf, np.arange(len(xf), dtype=np.uint32))]}
        t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
        assert t.get_option("stack16_in_use") == expect and t.get_option("blas_top_k") == (748 if expect else 310) - (len(xf) - 1)
        wb = o.world_bound
        rays = random_rays(rc, 200_000, n_tris, wb[:3], wb[3:])
        want = o.trace(rays, nthreads=8)
        for s16 in (1, 0):
            t.set_option("stack16", s16)
            assert_hits_equal(t.trace(rays), want, f"{n_tris} triangles, stack16 {s16}")
        t.free()
