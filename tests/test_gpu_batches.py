"""rc_trace_closest_device_batches / rc_trace_any_device_batches: several INDEPENDENT device batches in one call.  The reference's
batch API is one ray array per call (trace_rays(tlas, rays), ext/RaycoreMakieExt.jl:81-87); this entry point lets the caller say that batches
do not depend on each other, so that they overlap on the scene's auxiliary streams.  Each batch's hits must be exactly what a single
rc_trace_*_device call gives -- i.e. the oracle's, bit for bit -- whatever the number of batches, their sizes, the caller's stream, or a
hipGraph capture around the call."""
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
def world(rc, oracle):
    cfg = rc.scenes.config_c3(lattice=(4, 4, 2))
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    wb = t.world_bound()
    return cfg, t, o, wb


def upload(torch, a):
    return torch.from_numpy(a.view(np.uint8).reshape(-1)).cuda()


@pytest.mark.parametrize("sizes", [(300_000,), (250_000, 250_001), (1, 63, 64, 65, 100_003), (200_000,) * 7, (0, 5000, 0, 70_000)])
@pytest.mark.parametrize("mode", ["closest", "any"])
def test_batches_equal_single_calls_and_the_oracle(rc, world, sizes, mode):
    import torch
    cfg, t, o, wb = world
    batches = [random_rays(rc, max(k, 1), 9000 + 13 * i + k % 7, wb.p_min, wb.p_max)[:k] for i, k in enumerate(sizes)]
    d_r = [upload(torch, b) if len(b) else torch.empty(0, dtype=torch.uint8, device="cuda") for b in batches]
    d_h = [torch.full((len(b) * 32,), 0xAB, dtype=torch.uint8, device="cuda") for b in batches]
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        t.trace_device_batches([x.data_ptr() for x in d_r], [x.data_ptr() for x in d_h], [len(b) for b in batches], mode=mode, stream=st.cuda_stream)
        # the call is ONE operation on the caller's stream: work enqueued behind it sees every batch's hits
        got = [h.clone() for h in d_h]
    st.synchronize()
    for i, b in enumerate(batches):
        if len(b) == 0:
            continue
        want = o.trace(b, mode=mode, nthreads=8)
        g = got[i].cpu().numpy().view(rc.HIT_DT)
        if mode == "closest":
            assert_hits_equal(g, want, f"batch {i} of {sizes}")
        else:
            assert np.array_equal(g["hit"], want["hit"]), (i, sizes)
    t.wait_for_gpu()
    assert t.get_option("claim_drift") == 0


def test_null_stream_and_repeated_calls(rc, world):
    import torch
    cfg, t, o, wb = world
    batches = [random_rays(rc, 150_000 + 1000 * i, 700 + i, wb.p_min, wb.p_max) for i in range(5)]
    want = [o.trace(b, nthreads=8) for b in batches]
    d_r = [upload(torch, b) for b in batches]
    d_h = [torch.zeros(len(b) * 32, dtype=torch.uint8, device="cuda") for b in batches]
    torch.cuda.synchronize()
    for rep in range(6):   # the same batches again and again: each auxiliary stream's claim-order history learns its own batches
        for h in d_h:
            h.zero_()
        torch.cuda.synchronize()
        t.trace_device_batches([x.data_ptr() for x in d_r], [x.data_ptr() for x in d_h], [len(b) for b in batches])   # stream=None: the null stream
        torch.cuda.synchronize()
        for i in range(5):
            assert_hits_equal(d_h[i].cpu().numpy().view(rc.HIT_DT), want[i], f"rep {rep} batch {i}")
    assert t.get_option("claim_drift") == 0


def test_errors_leave_the_stream_usable(rc, world):
    import torch
    cfg, t, o, wb = world
    lib = rc.lib()
    b = random_rays(rc, 10_000, 1, wb.p_min, wb.p_max)
    d_r, d_h = upload(torch, b), torch.zeros(len(b) * 32, dtype=torch.uint8, device="cuda")
    rp, hp, nn = (C.c_void_p * 2)(d_r.data_ptr(), None), (C.c_void_p * 2)(d_h.data_ptr(), d_h.data_ptr()), (C.c_uint64 * 2)(len(b), len(b))
    assert lib.rc_trace_closest_device_batches(t._h, rp, hp, nn, 2, None) != 0 and b"batch 1" in lib.rc_last_error()
    assert lib.rc_trace_closest_device_batches(t._h, None, hp, nn, 2, None) != 0
    assert lib.rc_trace_closest_device_batches(t._h, rp, hp, nn, -1, None) != 0
    assert lib.rc_trace_closest_device_batches(None, rp, hp, nn, 2, None) != 0
    assert lib.rc_trace_any_device_batches(t._h, rp, hp, nn, 0, None) == 0          # nothing to do
    dirty = build_product(rc, cfg)
    dirty.push_instances(1, rc.scenes.IDENTITY3x4[None], np.zeros(1, np.uint32))
    rp[1] = d_r.data_ptr()
    assert lib.rc_trace_closest_device_batches(dirty._h, rp, hp, nn, 2, None) != 0 and b"rc_sync" in lib.rc_last_error()
    dirty.free()
    t.trace_device_batches([d_r.data_ptr()], [d_h.data_ptr()], [len(b)])
    torch.cuda.synchronize()
    assert_hits_equal(d_h.cpu().numpy().view(rc.HIT_DT), o.trace(b, nthreads=4), "after the error paths")


def test_capture_and_replay(rc, world):
    """On a capturing stream the auxiliary streams join the capture through the fork event and are joined back before the call returns:
    the whole call becomes one sub-graph; a replay traces all the batches again."""
    import torch
    cfg, t, o, wb = world
    batches = [random_rays(rc, 120_000 + 5 * i, 40 + i, wb.p_min, wb.p_max) for i in range(6)]
    want = [o.trace(b, nthreads=8) for b in batches]
    d_r = [upload(torch, b) for b in batches]
    d_h = [torch.zeros(len(b) * 32, dtype=torch.uint8, device="cuda") for b in batches]
    st = torch.cuda.Stream()
    held_before = t.get_option("release_captures")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        t.trace_device_batches([x.data_ptr() for x in d_r], [x.data_ptr() for x in d_h], [len(b) for b in batches], stream=torch.cuda.current_stream().cuda_stream)
    assert t.get_option("release_captures") == held_before + 6      # one capture slot per captured launch
    for rep in range(3):
        for h in d_h:
            h.zero_()
        g.replay()
        torch.cuda.synchronize()
        for i in range(6):
            assert_hits_equal(d_h[i].cpu().numpy().view(rc.HIT_DT), want[i], f"replay {rep} batch {i}")
    del g
    t.set_option("release_captures", 1)
    # and eagerly afterwards, on the same auxiliary streams
    for h in d_h:
        h.zero_()
    t.trace_device_batches([x.data_ptr() for x in d_r], [x.data_ptr() for x in d_h], [len(b) for b in batches], stream=st.cuda_stream)
    st.synchronize()
    for i in range(6):
        assert_hits_equal(d_h[i].cpu().numpy().view(rc.HIT_DT), want[i], f"eager after capture, batch {i}")
    assert t.get_option("claim_drift") == 0


def test_two_host_threads_each_with_its_own_stream(rc, world):
    """The fork / join events and the auxiliary streams are the scene's: calls from several host threads are enqueued one at a time (a mutex around the
    enqueue, microseconds) and each remains one operation on ITS caller's stream."""
    import threading
    import torch
    cfg, t, o, wb = world
    sets = []
    for k in range(2):
        batches = [random_rays(rc, 90_000 + 777 * (3 * k + i), 300 + 10 * k + i, wb.p_min, wb.p_max) for i in range(3)]
        sets.append((batches, [o.trace(b, nthreads=8) for b in batches], [upload(torch, b) for b in batches],
                     [torch.zeros(len(b) * 32, dtype=torch.uint8, device="cuda") for b in batches], torch.cuda.Stream()))
    torch.cuda.synchronize()
    errors = []

    def worker(k):
        try:
            batches, want, d_r, d_h, st = sets[k]
            for rep in range(5):
                for h in d_h:
                    h.zero_()
                torch.cuda.current_stream().synchronize()
                t.trace_device_batches([x.data_ptr() for x in d_r], [x.data_ptr() for x in d_h], [len(b) for b in batches], stream=st.cuda_stream)
                st.synchronize()
                for i in range(3):
                    assert_hits_equal(d_h[i].cpu().numpy().view(rc.HIT_DT), want[i], f"thread {k} rep {rep} batch {i}")
        except BaseException as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors
    t.wait_for_gpu()
    assert t.get_option("claim_drift") == 0
