"""rc_view_factor_totals*: the per-triangle totals of the view-factor job -- received[j] = column sum j, emitted[i] = row sum i of the
reference's N x N matrix (src/kernels.jl:74-104; the column sums are what docs/src/viewfactors_content.md:62-68 computes from it) --
accumulated without the matrix.  They must equal the sums of the ORACLE's matrix exactly, for every partition of the rays over replicas,
for metadata with gaps / duplicates / out-of-range ids, and at C5's full size the sums of the product's own matrix."""
import numpy as np
import pytest

from helpers import build_oracle, build_product
from test_gpu_view_factors_host import room_cfg

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rc():
    import raycore_jl_amd
    assert raycore_jl_amd.device_count() > 0, "no GPU visible: the product has no CPU fallback"
    return raycore_jl_amd


def sums(m):
    return m.sum(axis=0, dtype=np.uint64), m.sum(axis=1, dtype=np.uint64)


def test_totals_equal_the_oracle_matrix_sums(rc, oracle):
    cfg = room_cfg(rc)
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    want = o.view_factors(192, seed=77, nthreads=8)
    recv, emit = rc.view_factor_totals(t, 192, seed=77)
    w_recv, w_emit = sums(want)
    assert recv.dtype == np.uint64 and np.array_equal(recv, w_recv) and np.array_equal(emit, w_emit)
    assert int(recv.sum()) == int(emit.sum()) == int(want.sum(dtype=np.uint64)) > 0
    assert t.last_kernel_ms() > 0 and t.get_option("claim_drift") == 0
    # every kernel shape of the drivers (plain / top level in LDS / partial tops) gives the same vectors
    for k in (3, 5, 6, -1):
        t.set_option("kernel", k)
        r2, e2 = rc.view_factor_totals(t, 192, seed=77)
        assert np.array_equal(r2, w_recv) and np.array_equal(e2, w_emit), k
    t.set_option("kernel", -1)
    # rays partitioned over 2 and 3 replicas (one device: partial vectors added on the host) and ragged ray counts
    others = [build_product(rc, cfg) for _ in range(2)]
    for scenes in ([t, others[0]], [t, others[0], others[1]]):
        r2, e2 = rc.view_factor_totals_multi(scenes, 192, seed=77)
        assert np.array_equal(r2, w_recv) and np.array_equal(e2, w_emit), len(scenes)
    want1 = o.view_factors(1, seed=3, nthreads=8)   # fewer rays than replicas: empty shards
    r2, e2 = rc.view_factor_totals_multi([t] + others, 1, seed=3)
    assert np.array_equal(r2, sums(want1)[0]) and np.array_equal(e2, sums(want1)[1])
    with pytest.raises(rc.RaycoreError, match="same scene twice"):
        rc.view_factor_totals_multi([t, t], 8, seed=1)
    for s in others:
        s.free()
    t.free()


def test_totals_device_entry_point_accumulates_shards(rc, oracle):
    """rc_view_factor_totals_device: source x ray shards accumulated into device vectors on the caller's stream (the unit of a
    multi-process run); NULL for one of the vectors is allowed."""
    import torch
    from raycore_jl_amd._capi import check, lib, ptr
    cfg = room_cfg(rc)
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    n = t.n_primitives()
    want = o.view_factors(96, seed=11, nthreads=8)
    acc = torch.zeros(2 * n, dtype=torch.int64, device="cuda")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for (s0, s1) in ((0, n // 2), (n // 2, n)):
            for (r0, r1) in ((0, 31), (31, 96)):
                check(lib().rc_view_factor_totals_device(t._h, 96, 11, s0, s1, r0, r1, ptr(acc.data_ptr()), ptr(acc.data_ptr() + 8 * n), ptr(st.cuda_stream)))
    st.synchronize()
    got = acc.cpu().numpy().view(np.uint64)
    # sources are FLAT primitive indices: the vectors are indexed by metadata whatever the source order
    assert np.array_equal(got[:n], sums(want)[0]) and np.array_equal(got[n:], sums(want)[1])
    only = torch.zeros(n, dtype=torch.int64, device="cuda")
    check(lib().rc_view_factor_totals_device(t._h, 96, 11, 0, n, 0, 96, ptr(only.data_ptr()), None, None))
    torch.cuda.synchronize()
    assert np.array_equal(only.cpu().numpy().view(np.uint64), sums(want)[0])
    # two jobs at once on two streams, each into its own vectors: the kernel's private scratch counters are per stream
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    v1, v2 = torch.zeros(2 * n, dtype=torch.int64, device="cuda"), torch.zeros(2 * n, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    for rep in range(3):
        v1.zero_(); v2.zero_()
        torch.cuda.synchronize()
        check(lib().rc_view_factor_totals_device(t._h, 96, 11, 0, n, 0, 96, ptr(v1.data_ptr()), ptr(v1.data_ptr() + 8 * n), ptr(s1.cuda_stream)))
        check(lib().rc_view_factor_totals_device(t._h, 96, 11, 0, n, 0, 96, ptr(v2.data_ptr()), ptr(v2.data_ptr() + 8 * n), ptr(s2.cuda_stream)))
        torch.cuda.synchronize()
        for v in (v1, v2):
            g = v.cpu().numpy().view(np.uint64)
            assert np.array_equal(g[:n], sums(want)[0]) and np.array_equal(g[n:], sums(want)[1]), rep
    from raycore_jl_amd import distributed as rd
    r, e = rd.view_factor_totals_distributed(t, 96, 11)  # one rank: the whole job, no collective
    assert np.array_equal(r, sums(want)[0]) and np.array_equal(e, sums(want)[1])
    t.free()


def test_totals_with_metadata_gaps_duplicates_and_out_of_range(rc, oracle):
    def meta(n):
        m = np.arange(1, n + 1, dtype=np.uint32)
        m[5:40] = 7
        m[100:110] = 0
        m[150] = n + 50
        return m
    cfg = room_cfg(rc, meta)
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    want = o.view_factors(128, seed=5, nthreads=8)
    recv, emit = rc.view_factor_totals(t, 128, seed=5)
    assert np.array_equal(recv, sums(want)[0]) and np.array_equal(emit, sums(want)[1])
    assert emit[6] > 0 and not emit[8:39].any()
    other = build_product(rc, cfg)
    r2, e2 = rc.view_factor_totals_multi([t, other], 128, seed=5)
    assert np.array_equal(r2, recv) and np.array_equal(e2, emit)
    other.free(); t.free()


def test_totals_instanced_scene(rc, oracle):
    """Hits are counted by the metadata of the hit primitive whatever instance it sits in (src/kernels.jl:93-97)."""
    cfg = rc.scenes.config_c3(lon=8, bands=5, lattice=(2, 2, 2))
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    want = o.view_factors(64, seed=9, nthreads=8)
    recv, emit = rc.view_factor_totals(t, 64, seed=9)
    assert np.array_equal(recv, sums(want)[0]) and np.array_equal(emit, sums(want)[1]) and recv.sum() > 0
    t.free()


def test_totals_c5_full_size_match_the_matrix(rc):
    """BASELINE C5 at full size (50 028 triangles x 4096 rays = 204.9 M rays): the totals equal the column / row sums of the product's own
    device matrix (which tests/test_gpu_c5_and_claims.py holds against the oracle), and 2- and 3-way ray partitions change nothing."""
    import torch
    from raycore_jl_amd._capi import check, lib, ptr
    cfg = rc.scenes.config_c5()
    t = build_product(rc, cfg)
    n, rpt = t.n_primitives(), cfg["rays_per_triangle"]
    recv, emit = rc.view_factor_totals(t, rpt, seed=7)
    ms = t.last_kernel_ms()
    m = torch.zeros(n * n, dtype=torch.int32, device="cuda")
    check(lib().rc_view_factors_device(t._h, rpt, 7, 0, n, 0, rpt, ptr(m.data_ptr()), 1, n, 0, 0, None))  # Julia layout: [src + N * dst]
    torch.cuda.synchronize()
    mm = m.view(n, n)  # mm[dst, src]
    w_recv = mm.sum(dim=1, dtype=torch.int64).cpu().numpy().astype(np.uint64)
    w_emit = mm.sum(dim=0, dtype=torch.int64).cpu().numpy().astype(np.uint64)
    del m, mm
    torch.cuda.empty_cache()
    assert np.array_equal(recv, w_recv) and np.array_equal(emit, w_emit)
    assert int(recv.sum()) == int(emit.sum()) > 50_000_000
    assert 0 < ms < 200, ms
    others = [build_product(rc, cfg) for _ in range(2)]
    r3, e3 = rc.view_factor_totals_multi([t] + others, rpt, seed=7)
    assert np.array_equal(r3, recv) and np.array_equal(e3, emit)
    for s in others:
        s.free()
    t.free()
