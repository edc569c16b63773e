"""The multi-DEVICE branches on real hardware: everything here needs at least two visible GPUs and is skipped on the one-GPU box, so that
whichever node the suite lands on, the code that would produce a scaling curve has executed and been checked -- RC_VF_MODE_ROWS and
RC_VF_MODE_RAYS (multi-rank RCCL communicator, grouped in-place ncclReduce over xGMI, cross-device event waits) against the
single-device matrix, rc_view_factor_totals_multi (one ncclReduce of 2 N u64), the ray-shard trace entry points and get_illumination
(SURVEY.md 8e).  Oracle-independent where the one-device result is itself oracle-checked elsewhere in the suite; the small cases are
also held against the oracle directly."""
import numpy as np
import pytest

from helpers import assert_hits_equal, build_oracle, build_product, random_rays
from test_gpu_view_factors_host import room_cfg

# Nothing below has ever run (the builder's box has one GPU): a multi-rank RCCL call that never returns must end the run with a stack dump,
# not hang the suite (pytest-timeout's thread method exits the process; signals do not interrupt a thread blocked inside the library).
pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900, method="thread")]


def _n_dev():
    try:
        import raycore_jl_amd
        return raycore_jl_amd.device_count()
    except Exception:  # noqa: BLE001  (no library in a CPU-only collection run)
        return 0


needs2 = pytest.mark.skipif(_n_dev() < 2, reason="needs >= 2 GPUs (the RCCL multi-rank branch cannot run on one device)")


@pytest.fixture(scope="module")
def rc():
    import raycore_jl_amd
    return raycore_jl_amd


@needs2
def test_view_factors_rows_and_rays_on_distinct_devices(rc, oracle):
    cfg = room_cfg(rc)
    g = min(rc.device_count(), 8)
    scenes = [build_product(rc, cfg, device=d) for d in range(g)]
    o = build_oracle(oracle, cfg)
    want = o.view_factors(192, seed=77, nthreads=8)
    n = scenes[0].n_primitives()
    for k in sorted({2, g}):
        for s in scenes[:k]:
            s.set_option("vf_chunk_bytes", 4 * n * 24)  # several chunks: the reduce of chunk i overlaps the trace of chunk i + 1
        assert np.array_equal(rc.view_factors_multi(scenes[:k], 192, seed=77, mode="rows"), want), ("rows", k)
        assert np.array_equal(rc.view_factors_multi(scenes[:k], 192, seed=77, mode="rays"), want), ("rays", k)
        recv, emit = rc.view_factor_totals_multi(scenes[:k], 192, seed=77)
        assert np.array_equal(recv, want.sum(axis=0, dtype=np.uint64)) and np.array_equal(emit, want.sum(axis=1, dtype=np.uint64)), ("totals", k)
    # a second call reuses the cached communicators
    assert np.array_equal(rc.view_factors_multi(scenes[:2], 192, seed=77, mode="rays"), want)
    for s in scenes:
        s.free()


@needs2
def test_c5_totals_rccl_equal_one_device(rc):
    cfg = rc.scenes.config_c5()
    g = min(rc.device_count(), 8)
    scenes = [build_product(rc, cfg, device=d) for d in range(g)]
    rpt = cfg["rays_per_triangle"]
    r1, e1 = rc.view_factor_totals(scenes[0], rpt, seed=7)
    rg, eg = rc.view_factor_totals_multi(scenes, rpt, seed=7)
    assert np.array_equal(r1, rg) and np.array_equal(e1, eg) and int(rg.sum()) > 50_000_000
    for s in scenes:
        s.free()


@needs2
def test_trace_and_illumination_on_distinct_devices(rc, oracle):
    cfg = rc.scenes.config_c3(lattice=(4, 4, 2))
    g = min(rc.device_count(), 8)
    scenes = [build_product(rc, cfg, device=d) for d in range(g)]
    o = build_oracle(oracle, cfg)
    wb = scenes[0].world_bound()
    rays = random_rays(rc, 300_007, 12, wb.p_min, wb.p_max)
    want = o.trace(rays, nthreads=16)
    assert_hits_equal(rc.trace_multi(scenes, rays), want, f"{g} devices")
    got_any = rc.trace_multi(scenes, rays, mode="any")
    assert np.array_equal(got_any["hit"], o.trace(rays, mode="any", nthreads=16)["hit"])
    vd = np.array([0.3, 0.2, 1.0], np.float32)
    assert np.array_equal(rc.get_illumination_multi(scenes, vd, grid_size=300), o.get_illumination(vd, 300, nthreads=16))
    for s in scenes:
        s.free()


def test_skips_are_the_only_reason_nothing_ran(rc):
    """On the one-GPU box this file's hardware tests are skipped: say so in the report instead of passing silently."""
    n = rc.device_count()
    assert n >= 1
    if n < 2:
        pytest.skip(f"{n} GPU visible: the multi-device RCCL tests above were skipped; replicas on one device are covered by test_gpu_multi_device.py / test_gpu_view_factor_totals.py")
