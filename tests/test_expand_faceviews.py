"""expand_faceviews (GeometryBasics, as used by build_and_append_blas!, src/instanced-bvh.jl:581-590) restated on the host mirror: a
mesh whose attributes have their own index sets becomes a single-index mesh; merged vertices are numbered by first appearance."""
import numpy as np


def cube():
    p = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)], np.float32)        # 8 corners, index = 4x + 2y + z
    quads = [(0, 1, 3, 2, (-1, 0, 0)), (4, 6, 7, 5, (1, 0, 0)), (0, 4, 5, 1, (0, -1, 0)), (2, 3, 7, 6, (0, 1, 0)), (0, 2, 6, 4, (0, 0, -1)), (1, 5, 7, 3, (0, 0, 1))]
    pf, nf, normals = [], [], []
    for k, (a, b, c, d, n) in enumerate(quads):
        normals.append(n)
        pf += [(a, b, c), (a, c, d)]
        nf += [(k, k, k), (k, k, k)]
    return p, np.array(pf), np.array(normals, np.float32), np.array(nf)


def test_cube_with_per_face_normals_and_per_face_metadata():
    import raycore_jl_amd as rc
    p, pf, normals, nf = cube()
    meta = np.arange(1, 13, dtype=np.uint32)
    pos, faces, attr = rc.expand_faceviews(p, pf, normals=(normals, nf), face_meta=(meta, None))
    # per-face metadata makes every face's corners unique: 12 faces x 3 corners
    assert len(pos) == 36 and faces.shape == (12, 3) and faces.dtype == np.uint32
    assert np.array_equal(faces.reshape(-1), np.arange(36))                      # first-appearance numbering
    assert np.array_equal(pos[faces], p[pf])                                      # same triangles
    assert np.array_equal(attr["normals"][faces], normals[nf])
    assert np.array_equal(attr["face_meta"][faces[:, 0]], meta)
    # without the per-face metadata the two triangles of a cube side share two corners: 6 sides x 4 = 24 vertices
    pos2, faces2, attr2 = rc.expand_faceviews(p, pf, normals=(normals, nf))
    assert len(pos2) == 24 and np.array_equal(pos2[faces2], p[pf]) and np.array_equal(attr2["normals"][faces2], normals[nf])
    first_seen = []
    for t in faces2.reshape(-1):
        if t not in first_seen:
            first_seen.append(t)
    assert first_seen == list(range(24))
    # a mesh that already has one index set is returned unchanged (up to the first-appearance renumbering, identity here)
    pos3, faces3, _ = rc.expand_faceviews(p, pf)
    assert np.array_equal(pos3[faces3], p[pf]) and len(pos3) == 8
