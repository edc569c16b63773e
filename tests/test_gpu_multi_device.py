"""closest_hit / any_hit batches and get_illumination on several devices of one process, behind the C ABI (rays are
independent units -- replicas of the scene, contiguous ray shards, no collective; get_illumination = ray-grid shards + a sum of the
N-long histograms).  The GPU box has one device, so the replicas here all live on device 0 -- the sharding, the per-device threads and
the error paths are the same code a node with one scene per GPU runs.  Everything must equal the oracle bit for bit."""
import ctypes as C

import numpy as np
import pytest

from helpers import assert_hits_equal, build_oracle, build_product, random_rays

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rc():
    import raycore_jl_amd
    assert raycore_jl_amd.device_count() > 0, "no GPU visible: the product has no CPU fallback"
    return raycore_jl_amd


@pytest.fixture(scope="module")
def replicas(rc, oracle):
    cfg = rc.scenes.config_c3(lattice=(4, 4, 2))
    scenes = [build_product(rc, cfg) for _ in range(3)]
    o = build_oracle(oracle, cfg)
    return cfg, scenes, o, scenes[0].world_bound()


@pytest.mark.parametrize("n", [1, 63, 64, 65, 200, 100_003])
def test_ragged_batches(rc, replicas, n):
    """Shard boundaries are multiples of 64 rays: batches smaller than the number of replicas x 64 leave shards empty, n not a multiple
    of 64 leaves a ragged last shard."""
    cfg, scenes, o, wb = replicas
    rays = random_rays(rc, n, 4000 + n, wb.p_min, wb.p_max)
    want_c, want_a = o.trace(rays, nthreads=8), o.trace(rays, mode="any", nthreads=8)
    for k in (1, 2, 3):
        assert_hits_equal(rc.trace_multi(scenes[:k], rays), want_c, f"closest, {k} replicas, {n} rays")
        got = rc.trace_multi(scenes[:k], rays, mode="any")
        assert np.array_equal(got["hit"], want_a["hit"]), (k, n)
    for s in scenes:
        assert s.get_option("claim_drift") == 0


def test_large_batch_takes_the_pipelined_path_per_replica(rc, replicas):
    """7 M rays over two replicas: each shard is above the 3 Mi-ray threshold, so both replicas run the upload / trace / download
    pipeline at once, each on its own streams and threads.  `out` is reused and must be overwritten everywhere."""
    cfg, scenes, o, wb = replicas
    n = 7_000_000 + 17
    rays = random_rays(rc, n, 99, wb.p_min, wb.p_max)
    want = o.trace(rays, nthreads=16)
    out = np.empty(n, dtype=rc.HIT_DT)
    out.view(np.uint8)[:] = 0xAB
    got = rc.trace_multi(scenes[:2], rays, out=out)
    assert got is out
    assert_hits_equal(got, want, "two replicas, pipelined shards")
    assert scenes[0].last_kernel_ms() > 0 and scenes[1].last_kernel_ms() > 0
    single = scenes[2].trace(rays)
    assert_hits_equal(got, single, "same as one replica's own call")


def test_empty_batch_and_zero_sized_arguments(rc, replicas):
    cfg, scenes, o, wb = replicas
    got = rc.trace_multi(scenes[:2], np.empty(0, dtype=rc.RAY_DT))
    assert len(got) == 0


def test_illumination_shares(rc, replicas):
    """Each replica traces its share of the grid's rays; the partial histograms add up to the single-device histogram (and the oracle's)."""
    cfg, scenes, o, wb = replicas
    for viewdir, grid in (((0.3, 0.2, 1.0), 257), ((1.0, 0.0, 0.0), 64), ((0.0, -1.0, 0.2), 1)):
        want = o.get_illumination(viewdir, grid, nthreads=8)
        one = rc.get_illumination(scenes[0], viewdir, grid)
        assert np.array_equal(one, want)
        for k in (1, 2, 3):
            got = rc.get_illumination_multi(scenes[:k], viewdir, grid)
            assert got.dtype == np.float32 and np.array_equal(got, want), (viewdir, grid, k)
            assert float(got.sum()) <= grid * grid


def test_errors(rc, replicas):
    cfg, scenes, o, wb = replicas
    rays = random_rays(rc, 1000, 5, wb.p_min, wb.p_max)
    with pytest.raises(rc.RaycoreError, match="same scene twice"):
        rc.trace_multi([scenes[0], scenes[0]], rays)
    with pytest.raises(rc.RaycoreError, match="same scene twice"):
        rc.get_illumination_multi([scenes[1], scenes[1]], (0, 0, 1), 8)
    other = build_product(rc, rc.scenes.config_c3(lattice=(2, 2, 1)))
    with pytest.raises(rc.RaycoreError, match="same geometry"):
        rc.trace_multi([scenes[0], other], rays)
    with pytest.raises(rc.RaycoreError, match="same geometry"):
        rc.get_illumination_multi([scenes[0], other], (0, 0, 1), 8)
    # a replica with a pending mutation (through the raw entry point: the Python accel syncs on dispatch like Adapt.adapt): rc_sync first
    lib = rc.lib()
    hits = np.empty(len(rays), dtype=rc.HIT_DT)
    dirty = build_product(rc, cfg)
    dirty.push_instances(1, rc.scenes.IDENTITY3x4[None], np.zeros(1, np.uint32))
    handles = (C.c_void_p * 2)(scenes[0]._h, dirty._h)
    assert lib.rc_trace_closest_multi(handles, 2, rays.ctypes.data, hits.ctypes.data, len(rays)) != 0
    assert b"rc_sync" in lib.rc_last_error()
    # NULL scene list / NULL scene
    assert lib.rc_trace_closest_multi(None, 1, rays.ctypes.data, hits.ctypes.data, len(rays)) != 0
    handles = (C.c_void_p * 2)(scenes[0]._h, None)
    assert lib.rc_trace_any_multi(handles, 2, rays.ctypes.data, hits.ctypes.data, len(rays)) != 0
    assert lib.rc_trace_closest_multi(handles, 1, None, hits.ctypes.data, len(rays)) != 0   # NULL rays with n > 0
    # and the scenes are still usable
    assert_hits_equal(rc.trace_multi(scenes[:2], rays), o.trace(rays), "after the error paths")


def test_concurrent_multi_calls_on_the_same_replicas(rc, replicas):
    """Two host threads, each with its own batch, both naming the same two replicas (in opposite orders): the trace entry points are
    re-entrant per scene, so the calls interleave freely and both get their own hits."""
    import threading
    cfg, scenes, o, wb = replicas
    batches = [random_rays(rc, 300_000 + 77 * k, 60 + k, wb.p_min, wb.p_max) for k in range(2)]
    want = [o.trace(b, nthreads=8) for b in batches]
    errors = []

    def worker(k):
        try:
            order = scenes[:2] if k == 0 else scenes[1::-1]
            for it in range(6):
                assert_hits_equal(rc.trace_multi(order, batches[k]), want[k], f"thread {k} call {it}")
        except BaseException as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    for s in scenes:
        s.wait_for_gpu()
        assert s.get_option("claim_drift") == 0
