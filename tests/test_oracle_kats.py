"""Pins the CPU oracle against every known-answer test the reference's own suite holds for the
TLAS/BLAS path.  Each test cites the reference test it restates
(paths relative to /root/reference/)."""
import numpy as np
import pytest

UNIT_TRI = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0]], dtype=np.float32)


def xlat4(x, y, z):
    # Mat4f(1,0,0,0, 0,1,0,0, 0,0,1,0, x,y,z,1): column-major, translation in column 4
    return [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, x, y, z, 1]


def single_tri_scene(oracle, meta=1, xforms=(None,), tri=UNIT_TRI):
    s = oracle.Scene()
    b = s.add_blas(tri, meta=[meta])
    for i, x in enumerate(xforms):
        s.add_instance(b, None if x is None else oracle.mat4_to_mat3x4(x), instance_id=i + 1)
    return s.build()


def test_morton_ordering(oracle):  # test/test_instanced_bvh.jl:20-37
    L = oracle.lib()
    code = lambda p: L.rco_morton_code_30bit(np.array(p, dtype=np.float32).ctypes.data)
    c1, c2, c3 = code([0, 0, 0]), code([1, 1, 1]), code([0.5, 0.5, 0.5])
    assert c1 < c2 and c1 < c3 < c2


def test_bounds3_corner_order(oracle):  # test/bounds.jl:92-103 (corner(b, c): bit 0 -> x, bit 1 -> y, bit 2 -> z; src/bounds.jl:53-59)
    L = oracle.lib()
    mn, mx = np.zeros(3, np.float32), np.ones(3, np.float32)
    want = [(0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0), (0, 0, 1), (1, 0, 1), (0, 1, 1), (1, 1, 1)]
    for c, w in enumerate(want, start=1):
        out = np.empty(3, np.float32)
        L.rco_corner(mn.ctypes.data, mx.ctypes.data, c, out.ctypes.data)
        assert tuple(out) == w, (c, out)
    # the instance world AABB is the min / max over these eight corners through the forward transform (src/instanced-bvh-kernels.jl:38-62):
    # a quarter turn about z maps the unit cube [0,1]^3 to [-1,0] x [0,1] x [0,1]
    s = oracle.Scene()
    tri = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 1]], dtype=np.float32)           # root AABB = the unit cube
    s.add_blas(tri, np.array([1], np.uint32))
    rot = np.array([[0, -1, 0, 0], [1, 0, 0, 0], [0, 0, 1, 0]], dtype=np.float32)
    s.add_instance(1, rot, 0)
    st = s.build()
    wb = st.world_bound
    assert np.array_equal(wb[:3], np.float32([-1, 0, 0])) and np.array_equal(wb[3:], np.float32([0, 1, 1]))


def test_expand_bits_clz(oracle):  # test/test_instanced_bvh.jl:175-184
    L = oracle.lib()
    assert L.rco_expand_bits(0) == 0
    assert L.rco_expand_bits(0x3FF) == 0x09249249
    assert L.rco_clz32(0) == 32 and L.rco_clz32(1) == 31 and L.rco_clz32(0x80000000) == 0


def test_delta_oob(oracle):  # test/test_instanced_bvh.jl:186-200
    L = oracle.lib()
    codes = np.array([1, 2, 4, 8], dtype=np.uint32)
    assert L.rco_delta(1, 10, codes.ctypes.data, 4) == -1
    assert L.rco_delta(1, 2, codes.ctypes.data, 4) == 30   # clz(1^2)
    assert L.rco_delta(2, 3, codes.ctypes.data, 4) == 29
    same = np.array([5, 5], dtype=np.uint32)
    assert L.rco_delta(1, 2, same.ctypes.data, 2) == 32 + 30  # index tiebreak: clz(1^2)


def test_blas_single_triangle(oracle):  # test/test_instanced_bvh.jl:39-58
    s = single_tri_scene(oracle)
    n = s.blas_nodes
    assert len(n) == 1 and len(s.blas_prims) == 1
    assert n[0]["child0"] == oracle.INVALID_NODE and n[0]["child1"] == 1


def test_blas_two_triangles(oracle):  # test/test_instanced_bvh.jl:60-99
    quad = np.array([[0, 0, 0, 1, 0, 0, 1, 1, 0], [0, 0, 0, 1, 1, 0, 0, 1, 0]], dtype=np.float32)
    s = oracle.Scene()
    b = s.add_blas(quad)
    s.add_instance(b)
    s.build()
    n = s.blas_nodes
    assert len(n) == 3 and len(s.blas_prims) == 2
    assert n[0]["child0"] != oracle.INVALID_NODE  # root interior
    d = s.blas_descs[0]
    assert np.allclose(d["root_min"][:2], 0) and np.allclose(d["root_max"][:2], 1)
    # leaves: nodes n..2n-1, child1 = 1-based sorted primitive index
    assert {int(n[1]["child1"]), int(n[2]["child1"])} == {1, 2}
    assert n[1]["parent"] == 1 and n[2]["parent"] == 1


def test_transform_utils(oracle):  # test/test_instanced_bvh.jl:122-147
    L = oracle.lib()
    m = oracle.mat4_to_mat3x4(xlat4(5, 10, 15))
    p = np.array([1, 2, 3], dtype=np.float32)
    out = np.zeros(3, dtype=np.float32)
    L.rco_transform_point(m.ctypes.data, p.ctypes.data, out.ctypes.data)
    assert np.allclose(out, [6, 12, 18])
    v = np.array([1, 0, 0], dtype=np.float32)
    L.rco_transform_direction(m.ctypes.data, v.ctypes.data, out.ctypes.data)
    assert np.allclose(out, v)
    inv = oracle.mat3x4_inverse(m)
    assert np.allclose(inv, oracle.mat4_to_mat3x4(xlat4(-5, -10, -15)))


def test_mat3x4_inverse_general(oracle):
    rng = np.random.default_rng(1)
    for _ in range(20):
        A = rng.normal(size=(3, 3)).astype(np.float32) + 2 * np.eye(3, dtype=np.float32)
        t = rng.normal(size=3).astype(np.float32)
        m = np.concatenate([A, t[:, None]], axis=1).astype(np.float32).reshape(12)
        inv = oracle.mat3x4_inverse(m).reshape(3, 4)
        Ai = np.linalg.inv(A.astype(np.float64))
        assert np.allclose(inv[:, :3], Ai, rtol=1e-4, atol=1e-5)
        assert np.allclose(inv[:, 3], -Ai @ t, rtol=1e-4, atol=1e-5)


def test_tlas_single_instance(oracle):  # test/test_instanced_bvh.jl:206-227
    s = single_tri_scene(oracle)
    assert len(s.instances) == 1 and len(s.blas_descs) == 1 and len(s.tlas_nodes) == 1
    leaf = s.tlas_nodes[0]
    assert leaf["child0"] == oracle.INVALID_NODE and leaf["child1"] == 0


def test_tlas_two_instances(oracle):  # test/test_instanced_bvh.jl:229-268
    s = single_tri_scene(oracle, xforms=(None, xlat4(5, 0, 0)))
    assert len(s.instances) == 2 and len(s.blas_descs) == 1 and len(s.tlas_nodes) == 3
    wb = s.world_bound
    assert wb[0] == pytest.approx(0.0) and wb[3] == pytest.approx(6.0)


def test_closest_hit_basic(oracle):  # test/test_instanced_bvh.jl:274-302
    s = single_tri_scene(oracle, meta=42)
    h = s.trace(oracle.make_rays([[0.25, 0.25, 1.0], [2, 2, 1.0]], [0, 0, -1]))
    assert h[0]["hit"] == 1 and h[0]["t"] == pytest.approx(1.0)
    assert s.blas_prims[h[0]["primitive_id"]]["meta"] == 42
    assert h[1]["hit"] == 0


def test_closest_hit_translated(oracle):  # test/test_instanced_bvh.jl:304-339
    s = single_tri_scene(oracle, xforms=(xlat4(10, 0, 0),))
    h = s.trace(oracle.make_rays([[0.25, 0.25, 1.0], [10.25, 0.25, 1.0]], [0, 0, -1]))
    assert h[0]["hit"] == 0
    assert h[1]["hit"] == 1 and h[1]["t"] == pytest.approx(1.0)


def test_closest_hit_nearest_of_two(oracle):  # test/test_instanced_bvh.jl:341-378
    s = single_tri_scene(oracle, xforms=(None, xlat4(0, 0, -5)))
    h = s.trace(oracle.make_rays([[0.25, 0.25, 1.0]], [0, 0, -1]))[0]
    assert h["hit"] == 1 and h["t"] == pytest.approx(1.0)
    assert h["instance_id"] + 1 == 1  # reference returns the 1-based array position


def test_any_hit_basic(oracle):  # test/test_instanced_bvh.jl:380-405
    s = single_tri_scene(oracle)
    h = s.trace(oracle.make_rays([[0.25, 0.25, 1.0], [2, 2, 1.0]], [0, 0, -1]), mode="any")
    assert h[0]["hit"] == 1 and h[1]["hit"] == 0


def test_81_instances_161_nodes(oracle):  # test/test_instanced_bvh.jl:788-805
    xf = [xlat4(((i - 1) % 9) * 1.5, ((i - 1) // 9) * 1.25, 0) for i in range(1, 82)]
    s = single_tri_scene(oracle, xforms=xf)
    assert len(s.instances) == 81 and len(s.tlas_nodes) == 161


def test_kernel_batch_basic(oracle):  # test/test_instanced_bvh.jl:807-845
    s = single_tri_scene(oracle)
    o = [[0.25, 0.25, 1.0], [0.5, 0.25, 1.0], [5.0, 5.0, 1.0], [-1.0, -1.0, 1.0]]
    h = s.trace(oracle.make_rays(o, [0, 0, -1]))
    assert list(h["hit"]) == [1, 1, 0, 0]
    assert h["t"][0] == pytest.approx(1.0) and h["t"][1] == pytest.approx(1.0)


def test_kernel_batch_any_hit(oracle):  # test/test_instanced_bvh.jl:847-876
    s = single_tri_scene(oracle)
    o = [[0.25, 0.25, 1.0], [0.1, 0.1, 1.0], [5.0, 5.0, 1.0], [0.9, 0.9, 1.0]]
    h = s.trace(oracle.make_rays(o, [0, 0, -1]), mode="any")
    assert list(h["hit"]) == [1, 1, 0, 0]


def test_kernel_instance_ids(oracle):  # test/test_instanced_bvh.jl:878-916
    s = single_tri_scene(oracle, xforms=(None, xlat4(5, 0, 0), xlat4(0, 5, 0)))
    o = [[0.25, 0.25, 1.0], [5.25, 0.25, 1.0], [0.25, 5.25, 1.0]]
    h = s.trace(oracle.make_rays(o, [0, 0, -1]))
    assert all(h["hit"] == 1)
    assert list(h["instance_id"] + 1) == [1, 2, 3]


def test_kernel_barycentrics(oracle):  # test/test_instanced_bvh.jl:954-992
    s = single_tri_scene(oracle)
    o = [[0.25, 0.25, 1.0], [0.1, 0.1, 1.0], [0.5, 0.0, 1.0]]
    h = s.trace(oracle.make_rays(o, [0, 0, -1]))
    assert all(h["hit"] == 1)
    w = (np.float32(1) - h["bary_u"]) - h["bary_v"]
    assert w[0] == pytest.approx(0.5, abs=0.01) and h["bary_u"][0] == pytest.approx(0.25, abs=0.01)
    assert w[1] == pytest.approx(0.8, abs=0.01) and h["bary_u"][1] == pytest.approx(0.1, abs=0.01)
    assert w[2] == pytest.approx(0.5, abs=0.01) and h["bary_u"][2] == pytest.approx(0.5, abs=0.01)


def test_kernel_full_trace(oracle):  # test/test_instanced_bvh.jl:994-1042
    s = oracle.Scene()
    for off in ((0, 0, 0), (5, 0, 0)):
        b = s.add_blas(UNIT_TRI + np.tile(off, 3).astype(np.float32))
        s.add_instance(b)
    s.build()
    o = [[0.25, 0.25, 2.0], [5.25, 0.25, 3.0], [10.0, 10.0, 1.0]]
    h = s.trace(oracle.make_rays(o, [0, 0, -1]))
    assert list(h["hit"]) == [1, 1, 0]
    assert h["t"][0] == pytest.approx(2.0) and h["t"][1] == pytest.approx(3.0)
    assert list(h["instance_id"][:2] + 1) == [1, 2]


def test_kernel_metadata_three_meshes(oracle):  # test/test_instanced_bvh.jl:918-952
    s = oracle.Scene()
    for off in ((0, 0, 0), (5, 0, 0), (0, 5, 0)):
        s.add_instance(s.add_blas(UNIT_TRI + np.tile(off, 3).astype(np.float32), meta=[0]))  # "metadata from mesh is 0 by default"
    s.build()
    o = [[0.25, 0.25, 1.0], [5.25, 0.25, 1.0], [0.25, 5.25, 1.0], [10.0, 10.0, 1.0]]
    h = s.trace(oracle.make_rays(o, [0, 0, -1]))
    assert list(h["hit"]) == [1, 1, 1, 0]
    assert list(s.blas_prims["meta"][h["primitive_id"][:3]]) == [0, 0, 0]


def test_kernel_dynamic_transform(oracle):  # test/test_instanced_bvh.jl:1044-1088: the scene after update_transform!(x = 10) + sync!
    before, after = single_tri_scene(oracle), single_tri_scene(oracle, xforms=(xlat4(10, 0, 0),))
    at_origin, at_ten = oracle.make_rays([[0.25, 0.25, 1.0]], [0, 0, -1]), oracle.make_rays([[10.25, 0.25, 1.0]], [0, 0, -1])
    assert before.trace(at_origin)["hit"][0] == 1
    assert after.trace(at_origin)["hit"][0] == 0
    h = after.trace(at_ten)
    assert h["hit"][0] == 1 and h["t"][0] == pytest.approx(1.0)


def test_kernel_dynamic_add_instance(oracle):  # test/test_instanced_bvh.jl:1090-1131: before and after the second push! + sync!
    rays = oracle.make_rays([[0.25, 0.25, 1.0], [5.25, 0.25, 1.0]], [0, 0, -1])
    one = oracle.Scene()
    one.add_instance(one.add_blas(UNIT_TRI))
    assert list(one.build().trace(rays)["hit"]) == [1, 0]
    two = oracle.Scene()
    for off in ((0, 0, 0), (5, 0, 0)):
        two.add_instance(two.add_blas(UNIT_TRI + np.tile(off, 3).astype(np.float32)))
    assert list(two.build().trace(rays)["hit"]) == [1, 1]


def test_kernel_64_ray_batch(oracle):  # test/test_instanced_bvh.jl:1133-1159
    s = single_tri_scene(oracle)
    o = [[0.25 + 0.5 * (i % 8) / 7, 0.25 + 0.5 * ((i // 8) % 8) / 7, 1.0] for i in range(64)]
    h = s.trace(oracle.make_rays(o, [0, 0, -1]))
    assert 0 < h["hit"].sum() < 64


def test_tlas_meshes_diag(oracle):  # test/test_intersection.jl:57-78
    s = oracle.Scene()
    for i in range(4):
        tri = np.array([[0, 0, 0, 1, 0, 0, 1, 1, 0]], dtype=np.float32) + np.tile([i * 3, i * 3, 0], 3).astype(np.float32)
        b = s.add_blas(tri, meta=[i + 1])
        s.add_instance(b, instance_id=i + 1)
    s.build()
    h = s.trace(oracle.make_rays([[0.5, 0.5, -1]], [0, 0, 1]))[0]
    assert h["hit"] == 1


def test_tlas_meshes_in_a_row(oracle):  # test/test_intersection.jl:80-103
    s = oracle.Scene()
    for i, z in enumerate([0, 4, 8]):
        b = s.add_blas(np.array([[-1, -1, z, 1, -1, z, 0, 1, z]], dtype=np.float32), meta=[i + 1])
        s.add_instance(b, instance_id=i + 1)
    s.build()
    h = s.trace(oracle.make_rays([[0, 0, -2]], [0, 0, 1]))[0]
    assert h["hit"] == 1 and h["t"] == pytest.approx(2.0)
    assert s.blas_prims[h["primitive_id"]]["meta"] == 1


def test_miss_sentinel(oracle):  # test/test_intersection.jl:120-142
    s = single_tri_scene(oracle)
    h = s.trace(oracle.make_rays([[100, 100, 100]], [1, 0, 0]))[0]
    assert h["hit"] == 0 and h["t"] == 0 and h["bary_u"] == 0 and h["bary_v"] == 0
    assert h["primitive_id"] == 0xFFFFFFFF and h["instance_id"] == 0xFFFFFFFF


def test_empty_tlas_misses(oracle):  # test/test_tlas_stress.jl:808-831
    s = oracle.Scene().build()
    h = s.trace(oracle.make_rays([[0, 0, 5]], [0, 0, -1]))[0]
    assert h["hit"] == 0


def test_degenerate_filter(oracle):  # src/instanced-bvh.jl:573-577,599-601
    tris = np.array([[0, 0, 0, 1, 0, 0, 2, 0, 0], [0, 0, 0, 1, 0, 0, 0, 1, 0]], dtype=np.float32)
    s = oracle.Scene()
    b = s.add_blas(tris)
    s.add_instance(b)
    s.build()
    assert len(s.blas_prims) == 1 and s.blas_prims[0]["meta"] == 2  # default meta = face index before filtering
    with pytest.raises(ValueError):
        oracle.Scene().add_blas(tris[:1])


def test_philox_kat(oracle):
    # Random123 kat_vectors for philox4x32-10
    L = oracle.lib()
    def ph(ctr, key):
        c, k, o = np.array(ctr, dtype=np.uint32), np.array(key, dtype=np.uint32), np.zeros(4, dtype=np.uint32)
        L.rco_philox4x32_10(c.ctypes.data, k.ctypes.data, o.ctypes.data)
        return [int(x) for x in o]
    assert ph([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert ph([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert ph([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
