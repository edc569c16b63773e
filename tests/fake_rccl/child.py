"""Child process of tests/test_gpu_fake_rccl.py: runs ONE case of the multi-rank view_factors code paths on device 0, with G scenes standing in
for G RCCL ranks and tests/fake_rccl/libfake_rccl.so standing in for librccl.so (the parent sets RC_RCCL_LIBRARY, RC_ENABLE_DEBUG_HOOKS and
RC_DEBUG_RANKS_SHARE_DEVICE in this process's environment BEFORE the product library is loaded -- which is why every case is its own
process: the product resolves RCCL once per process).  Prints one JSON line {"ok": true, ...}; any assertion ends the process non-zero.
A call that never returns is ended by the parent's timeout after faulthandler has dumped every thread's stack."""
import ctypes as C
import faulthandler
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
ROOT = os.path.dirname(TESTS)
for p in (ROOT, TESTS):
    if p not in sys.path:
        sys.path.insert(0, p)

RPT, SEED = 191, 77  # 191 rays per triangle: no G in {2, 3, 8} divides it, every rank gets a different share


def stats(fake):
    out = (C.c_uint64 * 6)()
    fake.fake_rccl_stats(out)
    return dict(zip(("worlds", "calls", "collectives", "elements", "group_launches", "max_ranks"), [int(v) for v in out]))


def main():
    case, g = sys.argv[1], int(sys.argv[2])
    faulthandler.dump_traceback_later(int(os.environ.get("RC_CHILD_DUMP_AFTER", "240")), exit=True)
    import raycore_jl_amd as rc
    from oracle import pyoracle
    from helpers import build_oracle, build_product
    from test_gpu_view_factors_host import room_cfg

    assert rc.device_count() > 0, "no GPU visible: the product has no CPU fallback"
    fake = None
    if os.environ.get("RC_RCCL_LIBRARY", "").endswith("libfake_rccl.so"):
        fake = C.CDLL(os.environ["RC_RCCL_LIBRARY"], mode=C.RTLD_GLOBAL)  # the same mapping the product's dlopen returns
        fake.fake_rccl_stats.argtypes = [C.POINTER(C.c_uint64)]
        fake.fake_rccl_fail_after.argtypes = [C.c_int64]
    cfg = room_cfg(rc)
    scenes = [build_product(rc, cfg) for _ in range(g)]
    n = scenes[0].n_primitives()
    o = build_oracle(pyoracle, cfg)
    want = o.view_factors(RPT, seed=SEED, nthreads=8)
    want_recv, want_emit = want.sum(axis=0, dtype=np.uint64), want.sum(axis=1, dtype=np.uint64)
    rows_per_chunk = 24
    n_chunks = (n + rows_per_chunk - 1) // rows_per_chunk
    for s in scenes:
        s.set_option("vf_chunk_bytes", 4 * n * rows_per_chunk)
    info = {"case": case, "ranks": g, "n": n, "n_chunks": n_chunks}

    def rays_call():
        return rc.view_factors_multi(scenes, RPT, seed=SEED, mode="rays")

    if case == "rays":
        # multi_rays: per-chunk grouped ncclReduce on the communication streams + the copier thread + the cross-stream event fan-in
        out = np.full((n, n), 0xDEADBEEF, dtype=np.uint32, order="F")
        got = rc.view_factors_multi(scenes, RPT, seed=SEED, mode="rays", out=out)
        assert got is out and np.array_equal(got, want), "RAYS partition differs from the oracle's matrix"
        st = stats(fake)
        assert st == {"worlds": 1, "calls": n_chunks * g, "collectives": n_chunks, "elements": n * n, "group_launches": n_chunks, "max_ranks": g}, st
        assert np.array_equal(rays_call(), want)                       # cached communicator set
        assert stats(fake)["worlds"] == 1 and stats(fake)["collectives"] == 2 * n_chunks
        assert np.array_equal(rc.view_factors_multi(scenes, RPT, seed=SEED, mode="rows"), want)  # ROWS needs no collective
        assert stats(fake)["collectives"] == 2 * n_chunks
        one_chunk = [s.set_option("vf_chunk_bytes", 4 * n * n) for s in scenes]  # noqa: F841  the whole matrix in one reduce
        assert np.array_equal(rays_call(), want)
        assert stats(fake)["collectives"] == 2 * n_chunks + 1
        info["stats"] = stats(fake)
    elif case == "totals":
        # rc_view_factor_totals_multi: ONE ncclReduce(ncclUint64, 2 N) behind every rank's trace
        recv, emit = rc.view_factor_totals_multi(scenes, RPT, seed=SEED)
        assert np.array_equal(recv, want_recv) and np.array_equal(emit, want_emit), "totals differ from the oracle matrix's sums"
        st = stats(fake)
        assert st == {"worlds": 1, "calls": g, "collectives": 1, "elements": 2 * n, "group_launches": 1, "max_ranks": g}, st
        r1, e1 = rc.view_factor_totals(scenes[0], RPT, seed=SEED)
        assert np.array_equal(r1, recv) and np.array_equal(e1, emit)
        for k in range(2, g + 1):  # every sub-set of ranks is its own communicator set
            rk, ek = rc.view_factor_totals_multi(scenes[:k], RPT, seed=SEED)
            assert np.array_equal(rk, want_recv) and np.array_equal(ek, want_emit), k
        assert scenes[0].last_kernel_ms() > 0
        info["stats"] = stats(fake)
    elif case == "prepare":
        # rc_multi_prepare: communicator + streams + staging vectors + a warm-up collective, outside the timed call
        fixed = rc.multi_prepare(scenes)
        assert fixed["rccl_ranks"] == g, fixed
        st = stats(fake)
        assert st["worlds"] == 1 and st["collectives"] == 1 and st["elements"] == 2 * n and st["calls"] == g, st
        recv, emit = rc.view_factor_totals_multi(scenes, RPT, seed=SEED)
        assert np.array_equal(recv, want_recv) and np.array_equal(emit, want_emit)
        assert stats(fake)["worlds"] == 1 and stats(fake)["collectives"] == 2
        fixed2 = rc.multi_prepare(scenes)                              # idempotent; the communicator is reused
        assert fixed2["rccl_ranks"] == g and stats(fake)["worlds"] == 1
        info["prepare"] = fixed
    elif case == "status_word":
        # a traversal-stack overflow reported by rank g - 1 (the sticky status word, raised by the test hook): the call drains every
        # stream, raises, names the device -- and the same scenes work again afterwards
        assert np.array_equal(rays_call(), want)
        scenes[g - 1].set_option("debug_set_overflow", 1)
        try:
            rays_call()
            raise AssertionError("the overflow report of rank %d was lost" % (g - 1))
        except rc.RaycoreError as e:
            assert "overflow" in str(e), str(e)
        assert np.array_equal(rays_call(), want)
        scenes[g - 1].set_option("debug_set_overflow", 1)
        try:
            rc.view_factor_totals_multi(scenes, RPT, seed=SEED)
            raise AssertionError("the overflow report of rank %d was lost (totals)" % (g - 1))
        except rc.RaycoreError as e:
            assert "overflow" in str(e), str(e)
        recv, emit = rc.view_factor_totals_multi(scenes, RPT, seed=SEED)
        assert np.array_equal(recv, want_recv) and np.array_equal(emit, want_emit)
    elif case == "reduce_failure":
        # ncclReduce itself fails on rank 1 of the SECOND chunk's collective (chunk 0's traces, reduce and copy are in flight): the group is
        # closed, the copier thread is told to stop, every stream drains, the error carries RCCL's text -- no hang -- and the next call
        # builds a fresh communicator set
        assert np.array_equal(rays_call(), want)
        worlds = stats(fake)["worlds"]
        fake.fake_rccl_fail_after(g + 1)
        try:
            rays_call()
            raise AssertionError("an ncclReduce failure went unreported")
        except rc.RaycoreError as e:
            assert "ncclReduce failed" in str(e) and "injected" in str(e), str(e)
        fake.fake_rccl_fail_after(-1)
        assert np.array_equal(rays_call(), want)
        assert stats(fake)["worlds"] == worlds + 1, "the failed communicator set was reused"
        fake.fake_rccl_fail_after(0)  # totals: the very first rank's call fails
        try:
            rc.view_factor_totals_multi(scenes, RPT, seed=SEED)
            raise AssertionError("an ncclReduce failure went unreported (totals)")
        except rc.RaycoreError as e:
            assert "ncclReduce failed" in str(e), str(e)
        fake.fake_rccl_fail_after(-1)
        recv, emit = rc.view_factor_totals_multi(scenes, RPT, seed=SEED)
        assert np.array_equal(recv, want_recv) and np.array_equal(emit, want_emit)
    elif case == "hook_off":
        # without RC_DEBUG_RANKS_SHARE_DEVICE the product behaves as shipped: scenes on one device are replicas, not ranks
        try:
            rays_call()
            raise AssertionError("RC_VF_MODE_RAYS accepted several scenes on one device")
        except rc.RaycoreError as e:
            assert "DISTINCT device" in str(e), str(e)
        assert rc.multi_prepare(scenes)["rccl_ranks"] == 0
        recv, emit = rc.view_factor_totals_multi(scenes, RPT, seed=SEED)   # host sum of the partial vectors
        assert np.array_equal(recv, want_recv) and np.array_equal(emit, want_emit)
        if fake is not None:
            assert stats(fake)["calls"] == 0 and stats(fake)["worlds"] == 0
    elif case == "bad_library":
        try:
            rays_call()
            raise AssertionError("a missing RC_RCCL_LIBRARY was ignored")
        except rc.RaycoreError as e:
            assert "RC_RCCL_LIBRARY=" in str(e) and "could not be loaded" in str(e), str(e)
    else:
        raise SystemExit("unknown case " + case)
    for s in scenes:
        s.wait_for_gpu()
        assert s.get_option("claim_drift") == 0
        s.free()
    info["ok"] = True
    print(json.dumps(info))


if __name__ == "__main__":
    main()
