"""Oracle: mesh ingestion (build_and_append_blas! after the decomposition, src/instanced-bvh.jl:555-608), the full
Triangle{UInt32} record and the shading epilogue.  Pinned by the reference's own statements about build_triangle:
tangents NaN, default uv (0,0),(1,0),(1,1), metadata = face index before the degenerate filter or face_meta[first vertex]."""
import numpy as np


def grid_mesh(n, seed=0, with_uv=True):
    """(n+1)^2 vertex grid over [0,1]^2 with a bumpy z, 2 n^2 faces, analytic-ish normals."""
    g = np.random.default_rng(seed)
    xs, ys = np.meshgrid(np.linspace(0, 1, n + 1), np.linspace(0, 1, n + 1), indexing="ij")
    z = 0.1 * np.sin(6 * xs) * np.cos(5 * ys)
    verts = np.stack([xs, ys, z], -1).reshape(-1, 3).astype(np.float32)
    nrm = g.normal(size=verts.shape).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    uv = np.stack([xs, ys], -1).reshape(-1, 2).astype(np.float32) if with_uv else None
    idx = lambda i, j: i * (n + 1) + j
    faces = []
    for i in range(n):
        for j in range(n):
            faces += [[idx(i, j), idx(i + 1, j), idx(i + 1, j + 1)], [idx(i, j), idx(i + 1, j + 1), idx(i, j + 1)]]
    return verts, np.array(faces, np.uint32), nrm, uv


def test_mesh_equals_soup_geometry(oracle):
    v, f, nrm, uv = grid_mesh(12)
    a = oracle.Scene()
    a.add_instance(a.add_mesh(v, f, nrm, uv))
    a.build()
    b = oracle.Scene()
    b.add_instance(b.add_blas(v[f].reshape(-1, 9)))
    b.build()
    assert a.blas_nodes.tobytes() == b.blas_nodes.tobytes() and a.blas_prims.tobytes() == b.blas_prims.tobytes()


def test_full_triangle_fields(oracle):
    v, f, nrm, uv = grid_mesh(6)
    f[5] = [3, 3, 9]  # degenerate face: dropped, later faces keep their pre-filter index as metadata (:595)
    s = oracle.Scene()
    s.add_instance(s.add_mesh(v, f, nrm, uv))
    s.build()
    tris = s.triangles
    assert len(tris) == len(f) - 1 and 6 not in tris["metadata"] and set(tris["metadata"]) == set(range(1, len(f) + 1)) - {6}
    for t in tris:
        face = f[t["metadata"] - 1]
        assert np.array_equal(t["vertices"], v[face]) and np.array_equal(t["normals"], nrm[face]) and np.array_equal(t["uv"], uv[face])
        assert np.isnan(t["tangents"]).all()
    # per-vertex face_meta: metadata = face_meta[first vertex of the face]
    fm = np.arange(1000, 1000 + len(v), dtype=np.uint32)
    s2 = oracle.Scene()
    s2.add_instance(s2.add_mesh(v, f, nrm, None, fm))
    s2.build()
    t2 = s2.triangles
    assert np.all(t2["uv"] == np.array([[0, 0], [1, 0], [1, 1]], np.float32))  # default uv (:561-565)
    firsts = {tuple(map(tuple, v[face])): fm[face[0]] for face in f}
    for t in t2:
        assert t["metadata"] == firsts[tuple(map(tuple, t["vertices"]))]


def test_soup_triangles_get_geometric_normals(oracle):
    s = oracle.Scene()
    s.add_instance(s.add_blas(np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0]], np.float32), [7]))
    s.build()
    t = s.triangles[0]
    assert np.array_equal(t["normals"], np.tile([0, 0, 1], (3, 1))) and t["metadata"] == 7 and np.isnan(t["tangents"]).all()


def test_shading_attributes(oracle):
    v, f, nrm, uv = grid_mesh(8, seed=3)
    s = oracle.Scene()
    s.add_instance(s.add_mesh(v, f, nrm, uv))
    s.build()
    g = np.random.default_rng(4)
    o = np.c_[g.random((500, 2)), np.full(500, 2.0)].astype(np.float32)
    rays = oracle.make_rays(o, [[0, 0, -1]])
    hits = s.trace(rays)
    sn, suv = s.shading_attributes(hits)
    tris = s.triangles
    ok = hits["hit"] == 1
    assert ok.sum() > 300 and not sn[~ok].any() and not suv[~ok].any()
    for i in np.nonzero(ok)[0][:50]:
        t = tris[hits["primitive_id"][i]]
        u, w = hits["bary_u"][i], hits["bary_v"][i]
        b = np.array([(np.float32(1) - u) - w, u, w], np.float32)
        n = (t["normals"] * b[:, None]).sum(0)
        assert np.allclose(sn[i], n / np.linalg.norm(n), atol=1e-5)
        assert np.allclose(suv[i], (t["uv"] * b[:, None]).sum(0), atol=1e-6)
        # the interpolated uv of this mesh is the hit position's (x, y)
        assert np.allclose(suv[i], o[i, :2], atol=1e-4)
