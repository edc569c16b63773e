"""GPU: the batch-slot state machine of cost-ordered claiming (order_select / order_commit inside the trace kernels, k_order_scatter, and the host's
pause / cadence / rebuild logic in rc_cost_order_setup, raycore.jl_amd/csrc/rc_traverse.hip) against
its Python restatement (tests/order_model.py), word for word after every launch of scripted sequences.  Every launch's hits are
the oracle's as well -- the claim order never changes a result -- but that alone would not notice a state machine that never learns."""
import ctypes

import numpy as np
import pytest

import order_model as om
from helpers import assert_hits_equal, build_oracle, build_product

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rc():
    import raycore_jl_amd
    assert raycore_jl_amd.device_count() > 0, "no GPU visible: the product has no CPU fallback"
    return raycore_jl_amd


class Rig:
    def __init__(self, rc, oracle_mod, res=(1280, 800)):
        import torch
        self.rc, self.torch = rc, torch
        self.sc = rc.scenes
        self.cfg = self.sc.config_c3(lattice=(4, 4, 2))
        self.o = build_oracle(oracle_mod, self.cfg)
        self.centre = self.cfg["lattice_centre"]
        k = np.arange(400) + 0.5                           # a Fibonacci lattice on the sphere: every two positions >= 10 degrees apart, far outside the matching threshold
        z, phi = 1.0 - 2.0 * k / 400.0, np.pi * (1.0 + 5.0 ** 0.5) * k
        dirs = np.stack([np.sqrt(1.0 - z * z) * np.cos(phi), np.sqrt(1.0 - z * z) * np.sin(phi), z], axis=1)
        self.eyes = self.centre + 13.0 * dirs
        self.res = res
        self.cache = {}
        self.hip = ctypes.CDLL("libamdhip64.so")

    def eye(self, batch):
        if isinstance(batch, tuple):                      # ("m", f): frame f of the camera that starts at position 0
            return self.eyes[0] + np.array([0.02 * batch[1], 0.01 * batch[1], 0.0])
        return self.eyes[batch]

    def rays(self, batch, check):
        if batch not in self.cache:
            r = self.sc.pinhole_rays(self.res[0], self.res[1], self.eye(batch), self.centre, 45.0)
            self.cache[batch] = (self.torch.from_numpy(r.view(np.uint8).reshape(-1)).cuda(), self.o.trace(r, nthreads=16) if check else None, len(r))
            if len(self.cache) > 24:
                self.cache.pop(next(iter(self.cache)))
        return self.cache[batch]

    def header(self, t):
        h = self.torch.empty(48, dtype=self.torch.int32, device="cuda")
        self.hip.hipMemcpy(ctypes.c_void_p(h.data_ptr()), ctypes.c_void_p(t.get_option("debug_ctl_ptr")), ctypes.c_size_t(192), 3)
        self.torch.cuda.synchronize()
        return om.device_words(h.cpu().numpy().view(np.uint32))


@pytest.mark.parametrize("name", list(om.scripts()))
def test_device_follows_the_model(rc, oracle, name):
    import torch
    rig = Rig(rc, oracle)
    t = build_product(rc, rig.cfg)
    model = om.History()
    script = om.scripts()[name]
    for k, (batch, near) in enumerate(script):
        check = k < 24 or k >= len(script) - 8           # (the oracle pass of a long script's middle is skipped: those launches run in natural order anyway)
        d, want, n = rig.rays(batch, check)
        out = torch.zeros(n * 32, dtype=torch.uint8, device="cuda")
        t.trace_device(d.data_ptr(), out.data_ptr(), n)
        torch.cuda.synchronize()
        if want is not None:
            assert_hits_equal(out.cpu().numpy().view(rc.HIT_DT), want, f"{name}: launch {k + 1}")
        expect = model.launch(batch, near)
        got = rig.header(t)
        assert got == expect, f"{name}: launch {k + 1}\n device {got}\n model  {expect}"   # (a launch inside a pause moves nothing but skip_left)
    assert t.get_option("claim_drift") == 0
    t.free()


def test_shapes_and_streams_keep_separate_histories(rc, oracle):
    """a second batch size and a second stream each get a history of their own; coming back to the first finds it as it was left"""
    import torch
    rig = Rig(rc, oracle)
    t = build_product(rc, rig.cfg)
    d, want, n = rig.rays(0, True)
    out = torch.zeros(n * 32, dtype=torch.uint8, device="cuda")
    side = torch.cuda.Stream()
    models = {"full": om.History(), "short": om.History(), "side": om.History()}

    def go(key, count, stream=0):
        t.trace_device(d.data_ptr(), out.data_ptr(), count, stream=stream)
        torch.cuda.synchronize()
        assert_hits_equal(out.cpu().numpy().view(rc.HIT_DT)[:count], want[:count], key)
        expect = models[key].launch(0)
        assert rig.header(t) == expect, (key, rig.header(t), expect)
    for key, count, stream in (("full", n, 0), ("full", n, 0), ("short", n - 8192, 0), ("full", n, 0), ("side", n, side.cuda_stream), ("short", n - 8192, 0),
                               ("side", n, side.cuda_stream), ("full", n, 0), ("short", n - 8192, 0), ("side", n, side.cuda_stream), ("full", n, 0)):
        go(key, count, stream)
    t.free()


def test_a_caller_that_enqueues_far_ahead_still_gets_its_order(rc, oracle):
    """Round 6: the host half decides about the rebuild kernels from pinned words the device writes -- words a caller that never waits reads LATE.
    The pause count is therefore written together with the number of the launch that wrote it, and the host takes one launch off it for every
    launch enqueued since.  Ten never-repeating batches (the shape pauses for 64 launches), then 64 + 40 launches of ONE batch and not a single
    synchronisation in between: the batch must end up traced in an order learned from its own recordings (before the fix the host saw "paused" for the
    whole burst, nothing was ever rebuilt, and the repeated batch ran in natural order for as long as the caller did not wait -- bench.py's
    repeated-batch extra did exactly that)."""
    import torch
    rig = Rig(rc, oracle)
    t = build_product(rc, rig.cfg)
    bufs = [rig.rays(100 + k, False) for k in range(10)]
    d, want, n = rig.rays(3, True)
    out = torch.zeros(n * 32, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for b, _, nb in bufs:
        t.trace_device(b.data_ptr(), out.data_ptr(), nb)
    for _ in range(64 + 40):
        t.trace_device(d.data_ptr(), out.data_ptr(), n)
    torch.cuda.synchronize()
    h = rig.header(t)
    assert h["skip_left"] == 0 and h["fresh"] == 0 and h["gen"][h["sel"]] >= 30, h
    assert h["has_order"][h["sel"]] == 1 and h["order_valid"] == 1, f"the burst never got its recordings rebuilt into an order: {h}"
    assert_hits_equal(out.cpu().numpy().view(rc.HIT_DT), want, "last launch of the burst")
    assert t.get_option("claim_drift") == 0
    t.free()
