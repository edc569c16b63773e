"""GPU parity: rc_add_mesh (device-side expansion + filter + build), rc_export_triangles (136-byte Triangle records) and
rc_shading_attributes_device against the oracle, bit for bit."""
import numpy as np
import pytest

from helpers import assert_hits_equal
from test_oracle_mesh import grid_mesh

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rc():
    import raycore_jl_amd
    assert raycore_jl_amd.device_count() > 0, "no GPU visible: the product has no CPU fallback"
    return raycore_jl_amd


def both(rc, po, meshes, soups=()):
    t, s = rc.TLAS(), po.Scene()
    for (v, f, nrm, uv, fm) in meshes:
        t.push_mesh(v, f, nrm, uvs=uv, face_meta=fm)
        s.add_instance(s.add_mesh(v, f, nrm, uv, fm))
    for (soup, meta) in soups:
        t.push(soup, meta=meta)
        s.add_instance(s.add_blas(soup, meta))
    t.sync()
    s.build()
    return t, s


def test_triangles_identical(rc, oracle):
    v, f, nrm, uv = grid_mesh(40, seed=1)
    f[17] = [5, 5, 5]
    f[100] = [7, 8, 7]
    v2, f2, nrm2, _ = grid_mesh(9, seed=2, with_uv=False)
    fm2 = np.arange(500, 500 + len(v2), dtype=np.uint32)
    soup = np.array([[0, 0, 3, 1, 0, 3, 0, 1, 3], [0, 0, 4, 2, 0, 4, 0, 2, 4]], np.float32)
    t, s = both(rc, oracle, [(v, f, nrm, uv, None), (v2, f2, nrm2, None, fm2)], [(soup, [11, 12])])
    st = t.adapt()
    assert st.all_blas_nodes.tobytes() == s.blas_nodes.tobytes()
    got, want = st.all_blas_triangles, s.triangles
    assert len(got) == len(want) == len(f) - 2 + len(f2) + 2
    assert got.tobytes() == want.tobytes()  # NaN tangents included: same bit pattern


def test_closest_hit_returns_full_triangle(rc, oracle):
    v, f, nrm, uv = grid_mesh(10, seed=5)
    t, s = both(rc, oracle, [(v, f, nrm, uv, None)])
    hit, tri, dist, bary, inst = rc.closest_hit(t, rc.Ray((0.31, 0.47, 2.0), (0, 0, -1)))
    assert hit and inst == 1
    face = f[tri.metadata - 1]
    assert np.array_equal(tri.vertices, v[face]) and np.array_equal(tri.normals, nrm[face]) and np.array_equal(tri.uv, uv[face])
    assert np.isnan(tri.tangents).all()
    miss = rc.closest_hit(t, rc.Ray((5, 5, 2.0), (0, 0, -1)))
    assert not miss[0] and not miss[1].normals.any() and not miss[1].uv.any()  # empty_triangle (src/triangle_mesh.jl:49-57)


def test_shading_attributes_device(rc, oracle):
    import torch
    v, f, nrm, uv = grid_mesh(64, seed=6)
    t, s = both(rc, oracle, [(v, f, nrm, uv, None)], [(np.array([[0, 0, 1, 1, 0, 1, 0, 1, 1]], np.float32), [9])])
    g = np.random.default_rng(7)
    n = 100000
    o = np.c_[g.random((n, 2)) * 1.2 - 0.1, np.full(n, 2.0)].astype(np.float32)
    rays = rc.scenes.make_rays(o, np.tile([0, 0, -1], (n, 1)))
    hits = t.trace(rays)
    assert_hits_equal(hits, s.trace(rays, nthreads=4), "mesh closest")
    d_h = torch.from_numpy(hits.view(np.uint8).reshape(-1)).cuda()
    d_n = torch.full((n, 3), 7.0, dtype=torch.float32, device="cuda")
    d_uv = torch.full((n, 2), 7.0, dtype=torch.float32, device="cuda")
    t.shading_attributes_device(d_h.data_ptr(), n, d_n.data_ptr(), d_uv.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    wn, wuv = s.shading_attributes(hits)
    assert np.array_equal(d_n.cpu().numpy().view(np.uint32), wn.view(np.uint32))
    assert np.array_equal(d_uv.cpu().numpy().view(np.uint32), wuv.view(np.uint32))
    assert 0 < hits["hit"].sum() < n
    # mirror reflection rays from the same hits (reflect, src/math.jl:80)
    d_r = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
    d_o = torch.zeros(n * 32, dtype=torch.uint8, device="cuda")
    t.reflection_rays_device(d_r.data_ptr(), d_h.data_ptr(), n, d_o.data_ptr(), bias=0.01)
    torch.cuda.synchronize()
    assert d_o.cpu().numpy().tobytes() == s.reflection_rays(rays, hits, 0.01).tobytes()


def test_mesh_errors(rc):
    v, f, nrm, uv = grid_mesh(3)
    t = rc.TLAS()
    bad = f.copy()
    bad[0, 0] = len(v)
    with pytest.raises(rc.RaycoreError):
        t.add_mesh(v, bad, nrm)
    with pytest.raises(rc.RaycoreError):  # all faces degenerate => "Geometry has no valid triangles" (:601)
        t.add_mesh(v, np.zeros((4, 3), np.uint32), nrm)
    with pytest.raises(ValueError):
        t.add_mesh(v, f, nrm[:-1])


def test_mesh_survives_compaction_and_update(rc, oracle):
    v, f, nrm, uv = grid_mesh(6, seed=8)
    t = rc.TLAS()
    h1 = t.push(np.array([[0, 0, 5, 1, 0, 5, 0, 1, 5]], np.float32))
    h2 = t.push_mesh(v, f, nrm, uvs=uv)
    t.sync()
    t.delete(h1)
    t.sync()  # the mesh BLAS is renumbered by compact_instances!; its attributes must follow
    tris = t.adapt().all_blas_triangles
    assert len(tris) == len(f)
    for tr in tris[:20]:
        face = f[tr["metadata"] - 1]
        assert np.array_equal(tr["normals"], nrm[face]) and np.array_equal(tr["uv"], uv[face])


def test_scene_save_load_roundtrip(rc, oracle, tmp_path):
    """rc_scene_save / rc_scene_load: a loaded scene exports the same arrays, keeps its handles and traces bit-identically."""
    v, f, nrm, uv = grid_mesh(20, seed=9)
    soup = rc.scenes.random_triangles(500, 3, lo=-1, hi=1, edge=0.3)
    t = rc.TLAS()
    xf, _, _ = rc.scenes.lattice_transforms(2, 2, 1, 2.5, 4)
    h_mesh = t.push_mesh(v, f, nrm, transforms=xf[:3].reshape(3, 12), uvs=uv, instance_ids=[5, 6, 7])
    h_soup = t.push(soup, xf[3:4].reshape(1, 12), instance_id=9)
    h_gone = t.push(soup[:10])
    t.sync()
    t.delete(h_gone)
    path = tmp_path / "scene.rcs"
    t.save(path)
    u = rc.TLAS.load(path)
    assert not u.is_valid(h_gone) and u.is_valid(h_mesh) and u.n_instances(h_mesh) == 3 and u.n_instances(h_soup) == 1
    a, b = t.adapt(), u.adapt()
    for name in ("nodes", "instances", "all_blas_nodes", "all_blas_prims", "blas_descriptors", "all_blas_triangles"):
        assert getattr(a, name).tobytes() == getattr(b, name).tobytes(), name
    g = np.random.default_rng(1)
    rays = rc.scenes.make_rays(g.uniform(-4, 4, (20000, 3)), rc.scenes.normalize(g.normal(size=(20000, 3))))
    assert_hits_equal(u.trace(rays), t.trace(rays), "loaded scene")
    # the loaded scene is a live, mutable TLAS: handles work, new handle ids do not collide
    u.update_transforms(h_mesh, xf[:3].reshape(3, 12) * np.float32(1.0))
    h_new = u.push(soup[:5])
    assert h_new.id not in (h_mesh.id, h_soup.id, h_gone.id)
    u.sync()
    with pytest.raises(rc.RaycoreError):
        rc.TLAS.load(tmp_path / "missing.rcs")
    (tmp_path / "junk.rcs").write_bytes(b"not a scene file at all, definitely")
    with pytest.raises(rc.RaycoreError):
        rc.TLAS.load(tmp_path / "junk.rcs")
    # a file is untrusted input: indices the kernels would follow are checked before anything is uploaded
    data = bytearray(path.read_bytes())
    _, n_blas, n_inst, _, n_handles, _ = np.frombuffer(data, np.uint32, 6, 8)
    off = 32 + 12 * int(n_handles) + 108 * int(n_inst)            # first geometry: header (6 u32), root box (6 f32), primitives (40 B), nodes (64 B)
    n_prims = int(np.frombuffer(data, np.uint32, 1, off)[0])
    child0_of_root = off + 24 + 24 + 40 * n_prims + 48
    assert 1 <= int(np.frombuffer(data, np.uint32, 1, child0_of_root)[0]) <= 2 * n_prims - 1
    data[child0_of_root:child0_of_root + 4] = np.uint32(0x7FFFFFFF).tobytes()
    (tmp_path / "bad_child.rcs").write_bytes(bytes(data))
    with pytest.raises(rc.RaycoreError, match="child index"):
        rc.TLAS.load(tmp_path / "bad_child.rcs")
    # in-range links that do not form a tree would make a ray walk in circles for ever (a hung GPU): a node that points at itself,
    # at its parent (the root), or two parents for one node -- all refused before anything is uploaded
    good = bytearray(path.read_bytes())
    root_rec = off + 24 + 24 + 40 * n_prims
    c0 = int(np.frombuffer(good, np.uint32, 1, root_rec + 48)[0])
    c1 = int(np.frombuffer(good, np.uint32, 1, root_rec + 52)[0])
    inner = c0 if c0 < n_prims else c1               # an internal child of the root (node indices below n are internal)
    assert inner < n_prims
    inner_rec = root_rec + 64 * (inner - 1)
    for name, where, value in (("self", inner_rec + 48, inner), ("to_root", inner_rec + 52, 1), ("two_parents", root_rec + 52, c0)):
        data = bytearray(good)
        data[where:where + 4] = np.uint32(value).tobytes()
        (tmp_path / f"cycle_{name}.rcs").write_bytes(bytes(data))
        with pytest.raises(rc.RaycoreError, match="do not form a tree"):
            rc.TLAS.load(tmp_path / f"cycle_{name}.rcs")
    # the nodes are what rays meet, the primitives are what the entry cull's spheres come from -- a file in which they
    # disagree (a leaf with other vertices than its primitive, a box that is not its child's box) would let "entry_cull" change results
    leaf_rec = root_rec + 64 * (2 * n_prims - 2)     # the last node is a leaf
    for name, where, pattern in (("leaf_vertex", leaf_rec, "leaf node's vertices"), ("root_box", root_rec + 12, "child box")):
        data = bytearray(good)
        old = np.frombuffer(data, np.float32, 1, where)[0]
        data[where:where + 4] = np.float32(old + 0.25).tobytes()
        (tmp_path / f"forged_{name}.rcs").write_bytes(bytes(data))
        with pytest.raises(rc.RaycoreError, match=pattern):
            rc.TLAS.load(tmp_path / f"forged_{name}.rcs")
    # counts in a header are checked against the file's length before anything is allocated for them
    data = bytearray(good)
    data[off:off + 8] = np.array([0x10000000, 0x1FFFFFFF], np.uint32).tobytes()   # n_prims = 2^28, n_nodes = 2 n - 1: consistent, and absent
    (tmp_path / "huge.rcs").write_bytes(bytes(data))
    with pytest.raises(rc.RaycoreError, match="truncated"):
        rc.TLAS.load(tmp_path / "huge.rcs")


def test_metadata_per_face_on_shared_vertices(rc, oracle):
    """TLAS(items, metadata_fn) evaluates metadata_fn(mesh_idx, face_idx) per FACE (src/instanced-bvh.jl:2300-2306).  On a quad mesh the
    two triangles of a quad, (a, b, c) and (a, c, d), share their FIRST vertex, so a per-vertex metadata array (rc_add_mesh's face_meta,
    the push! path) cannot tell them apart -- the later face overwrites the earlier one's word (the Julia wrapper did exactly
    that).  rc_add_mesh_face_metadata carries one word per face: metadata_fn = (mi, fi) -> fi comes back as fi on every primitive."""
    k = 9
    gx, gy = np.meshgrid(np.arange(k + 1, dtype=np.float32), np.arange(k + 1, dtype=np.float32), indexing="ij")
    v = np.stack([gx.ravel(), gy.ravel(), np.zeros((k + 1) ** 2, np.float32)], axis=1)
    idx = lambda i, j: i * (k + 1) + j
    f = []
    for i in range(k):
        for j in range(k):
            a, b, c, d = idx(i, j), idx(i + 1, j), idx(i + 1, j + 1), idx(i, j + 1)
            f += [(a, b, c), (a, c, d)]               # both start at vertex a
    f = np.array(f, np.uint32)
    nrm = np.tile(np.array([0, 0, 1], np.float32), (len(v), 1))
    t = rc.TLAS()
    b = t.add_mesh(v, f, nrm, metadata_per_face=np.arange(1, len(f) + 1, dtype=np.uint32))
    t.push_instances(b)
    t.sync()
    prims = t.adapt().all_blas_prims
    assert sorted(prims["meta"].tolist()) == list(range(1, len(f) + 1))   # every face kept its own word
    for p in prims[:: 7]:                                                  # and it is the word of THAT face: same three vertices
        face = f[int(p["meta"]) - 1]
        assert np.array_equal(p["v"], v[face])
    # a ray into the middle of each triangle of quad (3, 4) reports that triangle's face index
    q = 2 * (3 * k + 4)
    for fi in (q, q + 1):
        c = v[f[fi]].mean(axis=0)
        hit, tri, dist, bary, inst = rc.closest_hit(t, rc.Ray(o=(c[0], c[1], 1.0), d=(0.0, 0.0, -1.0)))
        assert hit and int(tri.metadata) == fi + 1
    # the per-vertex form on the same mesh is what loses information: the quad's two faces read the same vertex
    words = np.zeros(len(v), np.uint32)
    for fi in range(len(f)):
        words[f[fi][0]] = fi + 1
    u = rc.TLAS()
    u.push_instances(u.add_mesh(v, f, nrm, face_meta=words))
    u.sync()
    assert len(set(u.adapt().all_blas_prims["meta"].tolist())) == len(f) // 2
    with pytest.raises(ValueError):
        t.add_mesh(v, f, nrm, face_meta=words, metadata_per_face=np.arange(len(f), dtype=np.uint32))
    t.free(); u.free()


def test_scene_save_load_keeps_the_lds_plan(rc, tmp_path):
    """A loaded scene rebuilds its traversal copy: the single-BLAS / large-top-level renumbering (kernels 5 / 6) is planned again and
    every kernel of the loaded scene agrees with the original."""
    sc = rc.scenes
    for n_inst in (40, 400):
        g = np.random.default_rng(n_inst)
        xf = np.tile(sc.IDENTITY3x4, (n_inst, 1)).astype(np.float32)
        xf[:, [3, 7, 11]] = g.uniform(-5, 5, size=(n_inst, 3))
        t = rc.TLAS()
        t.push(sc.random_triangles(3000, 21, lo=-0.5, hi=0.5, edge=0.1), xf)
        t.sync()
        path = tmp_path / f"scene{n_inst}.rcs"
        t.save(path)
        u = rc.TLAS.load(path)
        u.sync()
        assert (u.get_option("tlas_top_k"), u.get_option("blas_top_k")) == (t.get_option("tlas_top_k"), t.get_option("blas_top_k"))
        assert u.get_option("blas_top_k") > 0 and (u.get_option("tlas_top_k") > 0) == (n_inst > 256)
        rays = sc.make_rays(g.uniform(-7, 7, (60000, 3)), sc.normalize(g.normal(size=(60000, 3))))
        ref = t.trace(rays)
        assert ref["hit"].any()
        for kernel in (0, 3, 5, 6):
            u.set_option("kernel", kernel)
            assert_hits_equal(u.trace(rays), ref, f"{n_inst} instances, loaded scene, kernel {kernel}")
        t.free(); u.free()


def test_update_mesh_replaces_geometry(rc, oracle):
    v, f, nrm, uv = grid_mesh(8, seed=10)
    v2, f2, nrm2, uv2 = grid_mesh(5, seed=11)
    t = rc.TLAS()
    h = t.push_mesh(v, f, nrm, uvs=uv)
    t.sync()
    t.update_mesh(h, v2 + np.float32(0.25), f2, nrm2, uvs=uv2)
    s = oracle.Scene()
    s.add_instance(s.add_mesh(v2 + np.float32(0.25), f2, nrm2, uv2))
    s.build()
    st = t.adapt()
    assert st.all_blas_nodes.tobytes() == s.blas_nodes.tobytes() and st.all_blas_triangles.tobytes() == s.triangles.tobytes()


def test_face_view_mesh_through_expand_faceviews(rc, oracle):
    """A mesh with per-attribute index sets (cube: 8 positions, 6 per-face normals, one metadata value per face) goes through the host
    mirror's expand_faceviews (GeometryBasics' role in build_and_append_blas!, src/instanced-bvh.jl:581-590) and rc_add_mesh: the
    triangles come back with the face's normal on all three corners and the face's metadata, and match the oracle's mesh path."""
    from test_expand_faceviews import cube
    p, pf, normals, nf = cube()
    meta = np.arange(101, 113, dtype=np.uint32)
    pos, faces, attr = rc.expand_faceviews(p, pf, normals=(normals, nf), face_meta=(meta, None))
    t, s = both(rc, oracle, [(pos, faces, attr["normals"], None, attr["face_meta"])])
    got, want = t.adapt().all_blas_triangles, s.triangles
    assert got.tobytes() == want.tobytes() and len(got) == 12
    hit, tri, dist, bary, inst = rc.closest_hit(t, rc.Ray((0.3, 0.6, 5.0), (0, 0, -1)))   # the z = 1 side, from above
    assert hit and abs(dist - 4.0) < 1e-6 and tri.metadata in (111, 112)
    assert np.array_equal(tri.normals, np.tile(np.array([0, 0, 1], np.float32), (3, 1)))
