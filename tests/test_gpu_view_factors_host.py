"""view_factors into the caller's HOST matrix (src/kernels.jl:74-104 returns a host Matrix{UInt32}): the chunked single-device path, the
multi-device entry point rc_view_factors_multi in both partitions, and the row-block entry point of the multi-process driver -- all
must give the oracle's matrix bit for bit (Philox is keyed by (seed; ray index, source primitive), so no partition changes a count)."""
import ctypes as C

import numpy as np
import pytest

from helpers import build_oracle, build_product

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rc():
    import raycore_jl_amd
    assert raycore_jl_amd.device_count() > 0, "no GPU visible: the product has no CPU fallback"
    return raycore_jl_amd


def room_cfg(rc, meta=None):
    sc = rc.scenes
    verts = np.concatenate([sc.fan_sphere(12, 7, centre=(0, 0, 0), radius=0.5), sc.box_room((-1.5, -1.5, -1.5), (1.5, 1.5, 1.5), 2)])
    n = len(verts)
    m = np.arange(1, n + 1, dtype=np.uint32) if meta is None else meta(n)
    return {"blas": [(verts, m)], "instances": [(1, sc.IDENTITY3x4[None], np.zeros(1, np.uint32))]}


@pytest.fixture(scope="module")
def room(rc, oracle):
    cfg = room_cfg(rc)
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    want = o.view_factors(192, seed=77, nthreads=8)
    return cfg, t, o, want


def test_single_device_chunked(rc, room):
    """rc_view_factors: one chunk (the default block holds the whole small matrix), many chunks (three blocks in rotation, both compute
    streams), chunks of a single row; a reused `out` whose stale contents must be overwritten."""
    cfg, t, o, want = room
    n = t.n_primitives()
    assert np.array_equal(rc.view_factors(t, 192, seed=77), want)
    out = np.full((n, n), 0xDEADBEEF, dtype=np.uint32, order="F")
    for chunk_bytes in (4 * n * 7, 4 * n * 64, 4 * n, 192 << 20):
        t.set_option("vf_chunk_bytes", chunk_bytes)
        out[:] = 0xDEADBEEF
        got = rc.view_factors(t, 192, seed=77, out=out)
        assert got is out and np.array_equal(got, want), chunk_bytes
    assert t.last_kernel_ms() > 0
    assert t.get_option("claim_drift") == 0


def test_multi_rows_two_scenes_on_one_device(rc, room):
    """RC_VF_MODE_ROWS with two (and three) scenes -- here all on device 0; on a node, one per GPU -- each tracing its block of rows
    straight into the shared host matrix."""
    cfg, t, o, want = room
    others = [build_product(rc, cfg) for _ in range(2)]
    for scenes in ([t], [t, others[0]], [t, others[0], others[1]]):
        for s in scenes:
            s.set_option("vf_chunk_bytes", 4 * t.n_primitives() * 16)
        got = rc.view_factors_multi(scenes, 192, seed=77, mode="rows")
        assert np.array_equal(got, want), len(scenes)
    with pytest.raises(rc.RaycoreError, match="same scene twice"):
        rc.view_factors_multi([t, t], 192, seed=77)
    small = rc.TLAS(0)
    small.push(np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0]], np.float32))
    small.sync()
    with pytest.raises(rc.RaycoreError, match="same geometry"):
        rc.view_factors_multi([t, small], 192, seed=77)
    for s in others + [small]:
        s.free()
    t.set_option("vf_chunk_bytes", 192 << 20)


def test_multi_rays_rccl_one_device(rc, room):
    """RC_VF_MODE_RAYS on a single device: the RCCL path (communicator over one device, chunked ncclReduce on the communication stream,
    2-D copy-out from the root) runs for real; with one rank the sum is the rank's own accumulator.  Two scenes on one device cannot
    form a communicator: a clear error, not a hang."""
    cfg, t, o, want = room
    t.set_option("vf_chunk_bytes", 4 * t.n_primitives() * 40)
    got = rc.view_factors_multi([t], 192, seed=77, mode="rays")
    assert np.array_equal(got, want)
    t.set_option("vf_chunk_bytes", 192 << 20)
    other = build_product(rc, cfg)
    with pytest.raises(rc.RaycoreError, match="DISTINCT device"):
        rc.view_factors_multi([t, other], 192, seed=77, mode="rays")
    other.free()


def test_rows_host_blocks_and_leading_dimension(rc, room):
    """rc_view_factors_rows_host: row blocks written into a wider host matrix (ld > N) leave everything else untouched and assemble the
    whole matrix -- what each rank of the multi-process driver does with the shared-memory matrix."""
    from raycore_jl_amd._capi import check, lib
    cfg, t, o, want = room
    n, ld = t.n_primitives(), t.n_primitives() + 5
    buf = np.full((ld, n), 0xABABABAB, dtype=np.uint32, order="F")
    t.set_option("vf_chunk_bytes", 4 * n * 10)
    for r0, r1 in ((0, n // 3), (n // 3, n // 3 + 1), (n // 3 + 1, n)):
        check(lib().rc_view_factors_rows_host(t._h, 192, 77, r0, r1, buf.ctypes.data_as(C.c_void_p), ld))
    t.set_option("vf_chunk_bytes", 192 << 20)
    assert np.array_equal(buf[:n, :], want)
    assert np.all(buf[n:, :] == 0xABABABAB)
    from raycore_jl_amd import distributed as rd
    m = rd.view_factors_host_matrix(t, 192, 77)  # one rank: creates the shared-memory matrix, fills all rows, unlinks the file
    assert m.flags["F_CONTIGUOUS"] and np.array_equal(np.asarray(m), want)


def test_metadata_with_duplicates_and_gaps(rc, oracle):
    """Metadata that are not a permutation of 1..N (src/kernels.jl:85-97 indexes result[src_meta, hit_meta] whatever they are): rows of
    duplicated metadata accumulate, rows nobody owns stay zero, metadata outside 1..N are dropped -- in every chunking."""
    def meta(n):
        m = np.arange(1, n + 1, dtype=np.uint32)
        m[5:40] = 7            # many sources share row 6
        m[100:110] = 0         # outside 1..N: never counted, as source or as target
        m[150] = n + 50
        return m
    cfg = room_cfg(rc, meta)
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    want = o.view_factors(128, seed=5, nthreads=8)
    n = t.n_primitives()
    for chunk_bytes in (192 << 20, 4 * n * 9, 4 * n):
        t.set_option("vf_chunk_bytes", chunk_bytes)
        assert np.array_equal(rc.view_factors(t, 128, seed=5), want), chunk_bytes
    other = build_product(rc, cfg)
    assert np.array_equal(rc.view_factors_multi([t, other], 128, seed=5, mode="rows"), want)
    t.set_option("vf_chunk_bytes", 4 * n * 33)
    assert np.array_equal(rc.view_factors_multi([t], 128, seed=5, mode="rays"), want)
    assert want[6].sum() > 0 and not want[8:39].any()  # the duplicates' rows: one accumulates, the others are empty
    other.free(); t.free()
