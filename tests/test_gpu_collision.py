"""GPU parity for the collision broad phase (rc_collision.hip) against the oracle's restatement of src/collision.jl:
identical ContactPair array (content AND order), identical collide_instances_any answers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TRIS = np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0], [0, 0, 1, 1, 0, 1, 0, 1, 1]], np.float32)


@pytest.fixture(scope="module")
def rc():
    import raycore_jl_amd
    assert raycore_jl_amd.device_count() > 0, "no GPU visible: the product has no CPU fallback"
    return raycore_jl_amd


def both(rc, po, n, spread, seed, rot=False):
    g = np.random.default_rng(seed)
    xf = np.tile(po.IDENTITY, (n, 1))
    xf[:, [3, 7, 11]] = (g.random((n, 3)) * spread).astype(np.float32)
    if rot:
        a = g.random(n).astype(np.float32) * 6.28
        xf[:, 0], xf[:, 1], xf[:, 4], xf[:, 5] = np.cos(a), -np.sin(a), np.sin(a), np.cos(a)
    s = po.Scene()
    b = s.add_blas(TRIS)
    for i in range(n):
        s.add_instance(b, xf[i], i)
    s.build()
    t = rc.TLAS()
    h = t.push(TRIS, xf.reshape(n, 12), instance_ids=np.arange(n, dtype=np.uint32))
    return t, h, s


@pytest.mark.parametrize("n,spread,seed,rot", [(1, 1, 0, False), (2, 0.5, 1, False), (2, 50, 2, False), (300, 6, 3, True), (5000, 20, 4, True), (5000, 60, 5, False), (256, 0.5, 6, False)])
def test_contacts_identical(rc, oracle, n, spread, seed, rot):
    t, _, s = both(rc, oracle, n, spread, seed, rot)
    want, _ = s.collide_instances()
    res = rc.collide_instances(t)
    assert res.num_contacts == len(want)
    got = np.stack([res.contacts["instance_a"], res.contacts["instance_b"]], axis=1) if res.num_contacts else np.zeros((0, 2), np.uint32)
    assert np.array_equal(got, want)


def test_after_transform_update_and_delete(rc, oracle):
    t, h, s = both(rc, oracle, 64, 3, 11)
    n0 = rc.collide_instances(t).num_contacts
    far = np.tile(oracle.IDENTITY, (64, 1))
    far[:, 3] = np.arange(64) * 100.0
    t.update_transforms(h, far)
    assert rc.collide_instances(t).num_contacts == 0 and n0 > 0  # refit path feeds the broad phase
    h2 = t.push(TRIS, far[:1])
    res = rc.collide_instances(t)
    assert res.num_contacts == 1 and tuple(res.contacts[0]) == (1, 65)
    rc.collide_instances_any(t, h, h2)  # value follows the reference's leaf-position lookup (see test_any_matches_oracle)
    t.delete(h2)
    assert rc.collide_instances(t).num_contacts == 0
    with pytest.raises(rc.RaycoreError):
        rc.collide_instances_any(t, h, h2)


def test_any_matches_oracle(rc, oracle):
    g = np.random.default_rng(21)
    t = rc.TLAS()
    s = oracle.Scene()
    b = s.add_blas(TRIS)
    handles, ranges, k = [], [], 0
    for grp in range(6):
        m = int(g.integers(1, 4))
        xf = np.tile(oracle.IDENTITY, (m, 1))
        xf[:, [3, 7, 11]] = (g.random((m, 3)) * 4).astype(np.float32)
        handles.append(t.push(TRIS, xf) if grp == 0 else t.push_instances(1, xf))
        for x in xf:
            s.add_instance(b, x, 0)
        ranges.append((k, m))
        k += m
    s.build()
    for i in range(6):
        for j in range(6):
            assert rc.collide_instances_any(t, handles[i], handles[j]) == s.collide_instances_any(ranges[i], ranges[j])


def test_device_buffer_entry_point(rc, oracle):
    import ctypes as C
    import torch
    from raycore_jl_amd._capi import check, lib, ptr
    t, _, s = both(rc, oracle, 1000, 8, 31)
    want, _ = s.collide_instances()
    t.sync()
    d = torch.zeros(len(want) * 2 + 16, dtype=torch.int32, device="cuda")
    n = C.c_uint64(0)
    check(lib().rc_collide_instances_device(t._h, ptr(d.data_ptr()), len(want) + 8, C.byref(n), None))
    torch.cuda.synchronize()
    assert n.value == len(want)
    assert np.array_equal(d.cpu().numpy().view(np.uint32)[:2 * len(want)].reshape(-1, 2), want)
    assert lib().rc_collide_instances_device(t._h, ptr(d.data_ptr()), 1, C.byref(n), None) != 0  # capacity too small is an error
