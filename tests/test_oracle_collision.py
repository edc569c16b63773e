"""Oracle restatement of src/collision.jl: pair set against an all-pairs AABB test, and the reference's output order
(per Morton-sorted leaf, back-to-front inside the leaf's range).  The reference has no collision test: parity unpinned."""
import numpy as np
import pytest

TRIS = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0], [0, 0, 1, 1, 0, 1, 0, 1, 1]], np.float32)


def lattice(po, n, spread, seed):
    g = np.random.default_rng(seed)
    s = po.Scene()
    b = s.add_blas(TRIS)
    for i in range(n):
        x = po.IDENTITY.copy()
        x[[3, 7, 11]] = g.random(3) * spread
        s.add_instance(b, x, i)
    return s.build()


def brute_pairs(s):
    n = len(s.instances)
    leaves = s.tlas_nodes[n - 1:] if n > 1 else s.tlas_nodes
    box = {int(l["child1"]): (l["aabb0_min"], l["aabb0_max"]) for l in leaves}
    return {(a + 1, b + 1) for a in range(n) for b in range(a + 1, n)
            if np.all(box[a][1] >= box[b][0]) and np.all(box[a][0] <= box[b][1])}


@pytest.mark.parametrize("n,spread,seed", [(1, 1, 0), (2, 0.5, 1), (2, 50, 2), (40, 3, 3), (400, 9, 4), (400, 2, 5)])
def test_pairs_match_all_pairs(oracle, n, spread, seed):
    s = lattice(oracle, n, spread, seed)
    c, counts = s.collide_instances()
    assert len(c) == (counts[-1] if n else 0)
    assert len(set(map(tuple, c.tolist()))) == len(c) and np.all(c[:, 0] < c[:, 1])
    assert set(map(tuple, c.tolist())) == brute_pairs(s)


def test_order_is_per_sorted_leaf(oracle):
    s = lattice(oracle, 200, 4, 7)
    c, counts = s.collide_instances()
    n = 200
    leaves = s.tlas_nodes[n - 1:]
    start = 0
    for i, end in enumerate(counts):  # leaf i owns [start, end): all its pairs have instance_a = that leaf's instance
        assert np.all(c[start:end, 0] == leaves[i]["child1"] + 1)
        start = end


def test_any(oracle):
    s = lattice(oracle, 3, 0.2, 9)  # all three overlap
    assert s.collide_instances_any((0, 1), (1, 2))
    s2 = lattice(oracle, 2, 1e4, 10)
    c, _ = s2.collide_instances()
    assert s2.collide_instances_any((0, 1), (1, 1)) == (len(c) == 1)
    assert s2.collide_instances_any((0, 1), (0, 1))  # a box overlaps itself
