"""Trace entry points are re-entrant on a synced scene (the reference's own drivers call closest_hit on one adapted accel
from Threads.@threads, src/kernels.jl:64,82): several host threads calling rc_trace_closest (host buffers) and rc_trace_any_device /
rc_trace_closest_device (device buffers, a stream per thread) on ONE scene at once get the same bits as a single thread, the
self-resetting claim counters end at zero, and every thread reads back the duration of ITS last launch."""
import threading

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
def scene(rc, oracle):
    cfg = rc.scenes.config_c3(lattice=(4, 4, 2))
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    wb = t.world_bound()
    return t, o, wb


def test_eight_threads_on_one_scene(rc, scene):
    import torch
    t, o, wb = scene
    n_threads, n_calls = 8, 50
    batches = [random_rays(rc, 20_000 + 997 * k, 100 + k, wb.p_min, wb.p_max) for k in range(n_threads)]
    want_closest = [o.trace(b, nthreads=4) for b in batches]
    want_any = [o.trace(b, mode="any", nthreads=4) for b in batches]
    errors = []

    def worker(k):
        try:
            rays = batches[k]
            n = len(rays)
            st = torch.cuda.Stream()
            d_rays = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
            d_hits = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
            out = np.empty(n, dtype=rc.HIT_DT)
            for it in range(n_calls):
                if it % 3 == 0:    # host buffers: upload, trace, download inside the call
                    got = t.trace(rays, out=out)
                    assert_hits_equal(got, want_closest[k], f"thread {k} call {it} host closest")
                    ms = t.last_kernel_ms()
                    assert 0.0 < ms < 1e3, ms
                elif it % 3 == 1:  # device buffers, any_hit, this thread's stream
                    d_hits.zero_()
                    torch.cuda.current_stream().synchronize()
                    t.trace_device(d_rays.data_ptr(), d_hits.data_ptr(), n, mode="any", stream=st.cuda_stream)
                    st.synchronize()
                    got = d_hits.cpu().numpy().view(rc.HIT_DT)
                    assert np.array_equal(got["hit"], want_any[k]["hit"]), f"thread {k} call {it} device any"
                else:              # device buffers, closest_hit
                    t.trace_device(d_rays.data_ptr(), d_hits.data_ptr(), n, mode="closest", stream=st.cuda_stream)
                    st.synchronize()
                    assert_hits_equal(d_hits.cpu().numpy().view(rc.HIT_DT), want_closest[k], f"thread {k} call {it} device closest")
        except BaseException as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    t.wait_for_gpu()
    assert t.get_option("claim_drift") == 0


def test_host_buffer_calls_beyond_the_context_pool(rc, scene):
    """More concurrent host-buffer calls than staging contexts (4): the extra calls wait for a context and still return the right bits;
    kernels of every variant."""
    t, o, wb = scene
    n_threads = 10
    batches = [random_rays(rc, 5_000 + 131 * k, 300 + k, wb.p_min, wb.p_max) for k in range(n_threads)]
    want = [o.trace(b) for b in batches]
    errors = []

    def worker(k):
        try:
            for it in range(12):
                assert_hits_equal(t.trace(batches[k]), want[k], f"thread {k} call {it}")
        except BaseException as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    for kernel in (-1, 3, 5):
        t.set_option("kernel", kernel)
        threads = [threading.Thread(target=worker, args=(k,)) for k in range(n_threads)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        assert not errors, (kernel, errors)
    t.set_option("kernel", -1)
    t.wait_for_gpu()
    assert t.get_option("claim_drift") == 0


def test_each_thread_reads_its_own_launch_time(rc, scene):
    """rc_last_kernel_ms reports the calling thread's latest launch, not whichever launch another thread enqueued since."""
    import torch
    t, o, wb = scene
    small = random_rays(rc, 4_096, 7, wb.p_min, wb.p_max)
    big = random_rays(rc, 4_000_000, 8, wb.p_min, wb.p_max)
    d_big = torch.from_numpy(big.view(np.uint8).reshape(-1)).cuda()
    d_out = torch.empty(len(big) * 32, dtype=torch.uint8, device="cuda")
    t.trace(small)
    ms_small = t.last_kernel_ms()
    result = {}

    def other():
        st = torch.cuda.Stream()
        t.trace_device(d_big.data_ptr(), d_out.data_ptr(), len(big), stream=st.cuda_stream)
        st.synchronize()
        result["big"] = t.last_kernel_ms()

    th = threading.Thread(target=other)
    th.start()
    th.join()
    assert result["big"] > 1.5 * ms_small   # ~0.9 ms against ~0.1 ms; the margin only has to tell the two launches apart
    assert abs(t.last_kernel_ms() - ms_small) < 1e-6  # this thread's own launch, still


@pytest.mark.gpu
def test_recent_kernel_ms_reports_every_launch_of_a_back_to_back_run(rc, oracle):
    """rc_recent_kernel_ms: the durations of a run of launches enqueued without waiting, from the events each launch carried on its own kernel
    dispatch; they agree with a bracket of HIP events around the whole run (kernels back to back: sum of durations <= bracket)."""
    import torch
    sc = rc.scenes
    cfg = sc.config_c3(lattice=(4, 4, 2))
    t = build_product(rc, cfg)
    rays = sc.pinhole_rays(1280, 800, cfg["eye"], cfg["lattice_centre"], 45.0)
    d = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
    out = torch.zeros(len(rays) * 32, dtype=torch.uint8, device="cuda")
    for _ in range(6):
        t.trace_device(d.data_ptr(), out.data_ptr(), len(rays))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        t.trace_device(d.data_ptr(), out.data_ptr(), len(rays), stream=torch.cuda.current_stream().cuda_stream)
    e1.record()
    e1.synchronize()
    ms = t.recent_kernel_ms(20)
    assert len(ms) == 20 and all(0.01 < x < 50.0 for x in ms), ms
    assert abs(ms[-1] - t.last_kernel_ms()) < 1e-6
    bracket = e0.elapsed_time(e1)
    assert 0.7 * bracket < sum(ms) <= bracket * 1.001, (sum(ms), bracket)
    assert len(t.recent_kernel_ms(47)) == 26 and t.recent_kernel_ms(0) == []
    t.free()
