"""Oracle BVH4 (oracle/rc_oracle.c, restating src/bvh4.jl): structure invariants and agreement with the BVH2 path.

The reference has no test for bvh4.jl (parity unpinned), so the restatement is pinned by
hand-derived small cases of gather_children_bvh2 / collapse_bvh2_to_bvh4 and by properties: every primitive sits in
exactly one leaf, parents and child slots are consistent, child boxes equal the BVH2 subtree boxes, and
closest_hit4 returns the same t as the BVH2 closest_hit and as brute force.
"""
import numpy as np
import pytest

INVALID = 0xFFFFFFFF


def soup(n, seed, scale=0.1):
    g = np.random.default_rng(seed)
    c = g.random((n, 1, 3)).astype(np.float32)
    e = (g.random((n, 3, 3)).astype(np.float32) - 0.5) * np.float32(scale)
    return (c + e).reshape(n, 9)


def scene_of(po, verts):
    s = po.Scene()
    b = s.add_blas(verts)
    s.add_instance(b)
    return s.build(), b


def test_single_triangle_is_one_leaf(oracle):  # src/bvh4.jl:334-351
    s, b = scene_of(oracle, np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0]], np.float32))
    n4 = s.blas4_nodes(b)
    assert len(n4) == 1
    nd = n4[0]
    assert nd["child_count"] == 0 and nd["primitive_count"] == 1 and nd["parent"] == INVALID
    assert list(nd["child"]) == [1, INVALID, INVALID, INVALID]
    assert np.array_equal(nd["aabb"][0], [[0, 0, 0], [1, 1, 0]]) and not nd["aabb"][1:].any()
    h = s.trace4(b, oracle.make_rays([[0.25, 0.25, 1]], [[0, 0, -1]]))
    assert h["hit"][0] == 1 and h["t"][0] == 1.0 and h["primitive_id"][0] == 0


def test_two_triangles_hand_derived(oracle):
    # BVH2: root(1) -> leaves 2,3.  gather: queue=[2,3], no interior => child 1 = node 2, child 2 = node 3 (:234-277);
    # numbering: root 1, then its leaves in slot order (:363-391).
    v = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0], [5, 0, 0, 6, 0, 0, 5, 1, 0]], np.float32)
    s, b = scene_of(oracle, v)
    n2, n4 = s.blas_nodes, s.blas4_nodes(b)
    assert len(n4) == 3
    root = n4[0]
    assert root["child_count"] == 2 and list(root["child"]) == [2, 3, INVALID, INVALID] and root["parent"] == INVALID
    assert n4[1]["child"][0] == n2[n2[0]["child0"] - 1]["child1"] and n4[2]["child"][0] == n2[n2[0]["child1"] - 1]["child1"]
    assert n4[1]["parent"] == 1 and n4[2]["parent"] == 1
    assert np.all(root["aabb"][2:, 0] == np.inf) and np.all(root["aabb"][2:, 1] == -np.inf)  # Bounds3() in unused slots


def test_four_triangles_full_node(oracle):
    # a balanced 4-leaf BVH2 collapses into one 4-wide root: both interior children get expanded (:241, :255)
    xs = [0.0, 1.0, 10.0, 11.0]
    v = np.array([[x, 0, 0, x + 0.5, 0, 0, x, 0.5, 0] for x in xs], np.float32)
    s, b = scene_of(oracle, v)
    n4 = s.blas4_nodes(b)
    n2 = s.blas_nodes
    if n2[n2[0]["child0"] - 1]["child0"] != INVALID and n2[n2[0]["child1"] - 1]["child0"] != INVALID:  # balanced split
        assert len(n4) == 5 and n4[0]["child_count"] == 4
        # expansion order: queue [A,B] -> expand A: [B,A0,A1] -> expand B: [A1,A0,B0,B1] -> pops 1st each time with the
        # last element moved into the hole: A1, B1, B0, A0
        a, bb = n2[0]["child0"], n2[0]["child1"]
        want = [n2[a - 1]["child1"], n2[bb - 1]["child1"], n2[bb - 1]["child0"], n2[a - 1]["child0"]]
        got_prims = [n4[c - 1]["child"][0] for c in n4[0]["child"]]
        assert got_prims == [n2[w - 1]["child1"] for w in want]


@pytest.mark.parametrize("n,seed", [(3, 1), (7, 2), (64, 3), (1000, 4), (5000, 5)])
def test_structure_invariants(oracle, n, seed):
    s, b = scene_of(oracle, soup(n, seed))
    n4 = s.blas4_nodes(b)
    leaves = n4["child_count"] == 0
    assert leaves.sum() == n and sorted(n4["child"][leaves, 0]) == list(range(1, n + 1))
    assert n4[0]["parent"] == INVALID
    prims = s.blas_prims
    for i, nd in enumerate(n4):
        if nd["child_count"] == 0:
            tri = prims[nd["child"][0] - 1]["v"]
            assert np.array_equal(nd["aabb"][0, 0], tri.min(0)) and np.array_equal(nd["aabb"][0, 1], tri.max(0))
            continue
        cc = nd["child_count"]
        assert 2 <= cc <= 4 and np.all(nd["child"][:cc] != INVALID) and np.all(nd["child"][cc:] == INVALID)
        for k in range(cc):
            ch = n4[nd["child"][k] - 1]
            assert ch["parent"] == i + 1
            if ch["child_count"] == 0:
                assert np.array_equal(nd["aabb"][k], ch["aabb"][0])
            else:  # slot box = union of the child's own child boxes
                c2 = ch["child_count"]
                assert np.array_equal(nd["aabb"][k, 0], ch["aabb"][:c2, 0].min(0)) and np.array_equal(nd["aabb"][k, 1], ch["aabb"][:c2, 1].max(0))


@pytest.mark.parametrize("n,seed", [(200, 11), (5000, 12)])
def test_closest4_agrees_with_bvh2_and_brute(oracle, n, seed):
    s, b = scene_of(oracle, soup(n, seed, 0.4 if n < 1000 else 0.1))
    g = np.random.default_rng(seed + 100)
    o = (g.random((3000, 3)) * 2 - 0.5).astype(np.float32)
    d = g.standard_normal((3000, 3)).astype(np.float32)
    rays = oracle.make_rays(o, d)
    h2, h4 = s.trace(rays, nthreads=4), s.trace4(b, rays, nthreads=4)
    assert np.array_equal(h2["hit"], h4["hit"]) and h4["hit"].sum() > 100
    assert np.array_equal(h2["t"].view(np.uint32), h4["t"].view(np.uint32))  # identity instance => same arithmetic per triangle
    same = h2["primitive_id"] == h4["primitive_id"]
    assert same.mean() > 0.999  # only exact-t ties between two triangles may resolve differently
    br = s.brute(rays[:300])
    assert np.array_equal(br["hit"], h4["hit"][:300]) and np.array_equal(br["t"], h4["t"][:300])
    a2, a4 = s.trace(rays, mode="any", nthreads=4), s.trace4(b, rays, mode="any", nthreads=4)
    assert np.array_equal(a2["hit"], a4["hit"])
    assert np.all(h4["instance_id"] == INVALID) and np.all(h4["instance_custom_index"] == 0)


def test_trace4_ignores_tmin(oracle):  # :610 ray_mint = 0, not ray.t_min
    s, b = scene_of(oracle, np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0]], np.float32))
    r = oracle.make_rays([[0.25, 0.25, 1]], [[0, 0, -1]], tmin=5.0)
    assert s.trace4(b, r)["hit"][0] == 1 and s.trace(r)["hit"][0] == 0
