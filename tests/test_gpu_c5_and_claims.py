"""GPU tests added in round 2: BASELINE config C5 at its stated size, and the work-claim machinery of the persistent kernels
(chunk counters that are never reset, shard count bounded by the waves of the launch, slot reuse across streams)."""
import numpy as np
import pytest

from helpers import assert_hits_equal, build_oracle, build_product, random_rays

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rc():
    import raycore_jl_amd
    assert raycore_jl_amd.device_count() > 0, "no GPU visible: the product has no CPU fallback"
    return raycore_jl_amd


def test_c5_full_size_view_factors(rc, oracle):
    """BASELINE C5 as stated: view_factors on the ~50 k-triangle closed scene with rays_per_triangle = 4096 (204.9 M rays, a
    10 GB N x N matrix), through rc_view_factors_device in all three partitions on one GPU (src/kernels.jl:74-104).
    Checked: 64 random source rows against the oracle's rows for the same (seed; source, ray) Philox keys, diag == 0
    (hit_meta != src_meta, :94), row sums <= R, and the three partitions produce the same matrix."""
    import torch
    from raycore_jl_amd import distributed as rd
    cfg = rc.scenes.config_c5()
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    n, rpt, seed = t.n_primitives(), cfg["rays_per_triangle"], 20260202
    assert n >= 50_000 and rpt == 4096
    meta = t._prims()["meta"].astype(np.int64)
    assert np.array_equal(np.sort(meta), np.arange(1, n + 1))  # metadata = 1..N: rows are a permutation of the sorted primitives

    # rows_sharded on one rank: block row r = source primitive r (Morton-sorted order), matrix row meta[r] - 1
    block, row_index = rd.view_factors_distributed(t, rpt, seed, mode="rows_sharded")
    torch.cuda.synchronize()
    assert block.shape == (n, n) and np.array_equal(row_index, meta - 1)
    g = np.random.default_rng(5)
    for src in g.choice(n, 64, replace=False):
        want = o.view_factor_row(rpt, int(src), seed=seed)
        got = block[int(src)].cpu().numpy().view(np.uint32)
        assert np.array_equal(got, want), f"source primitive {src}: {int((got != want).sum())} entries differ"
        assert got[meta[src] - 1] == 0 and got.sum() <= rpt
    row_sums = block.sum(dim=1, dtype=torch.int64)
    assert int(row_sums.max()) <= rpt and int(row_sums.min()) >= 0
    total = int(row_sums.sum())
    assert total > 0.5 * n * rpt  # a closed room: most rays hit something that is not their source
    diag = block[torch.arange(n, device=block.device), torch.as_tensor(meta - 1, device=block.device)]
    assert int(diag.abs().sum()) == 0
    # the matrix in metadata order, kept as a checksum per row to compare the other partitions without holding two 10 GB matrices
    w = torch.arange(1, n + 1, device=block.device, dtype=torch.int64)
    perm = torch.as_tensor(np.argsort(meta), device=block.device)   # matrix row m comes from block row perm[m]
    want_sum = row_sums[perm].cpu().numpy()
    want_chk = (block.to(torch.int64) * w).sum(dim=1)[perm].cpu().numpy()
    del block, diag, row_sums
    torch.cuda.empty_cache()
    for mode in ("rows", "rays"):
        m = rd.view_factors_distributed(t, rpt, seed, mode=mode)
        torch.cuda.synchronize()
        assert m.shape == (n, n)
        assert np.array_equal(m.sum(dim=1, dtype=torch.int64).cpu().numpy(), want_sum), mode
        assert np.array_equal((m.to(torch.int64) * w).sum(dim=1).cpu().numpy(), want_chk), mode
        del m
        torch.cuda.empty_cache()
    assert t.get_option("claim_drift") == 0
    t.free()


@pytest.mark.parametrize("pool", [16, 32, 64, 100])
def test_small_pools_trace_every_ray(rc, oracle, pool):
    """A launch with fewer waves than chunk counters must still trace every chunk (with 16 counters and 4 waves,
    chunks dealt to the 12 unpopulated shards were never claimed).  Sweeps small claim sizes over every persistent kernel,
    the BVH4 kernel and the drivers on batches of one to a few workgroups."""
    sc = rc.scenes
    xf, _, _ = sc.lattice_transforms(2, 2, 1, 1.3, 3)
    cfg = {"blas": [(sc.fan_sphere(16, 9), None)], "instances": [(1, xf, np.arange(len(xf), dtype=np.uint32))]}
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    t.set_option("pool", pool)
    wb = o.world_bound
    for n in (1, 63, 256, 300, 1000, 5000):
        rays = random_rays(rc, n, 100 + n, wb[:3], wb[3:])
        want, want_any = o.trace(rays), o.trace(rays, mode="any")
        for k in (1, 2, 3, 4, 5, 6):
            t.set_option("kernel", k)
            out = np.full(n, 0xAB, dtype=np.uint8).repeat(32).view(rc.HIT_DT)  # poisoned: an untraced ray cannot look like a result
            assert_hits_equal(t.trace(rays, out=out), want, f"pool {pool} n {n} kernel {k}")
            assert_hits_equal(t.trace(rays, mode="any"), want_any, f"pool {pool} n {n} kernel {k} any")
    t.set_option("kernel", -1)
    assert np.array_equal(rc.get_illumination(t, [0.2, 0.3, 1.0], 20), o.get_illumination([0.2, 0.3, 1.0], 20))
    t.free()
    # drivers + BVH4 on a single-instance scene
    verts = np.concatenate([sc.fan_sphere(8, 5, radius=0.4), sc.box_room((-1, -1, -1), (1, 1, 1), 1)])
    cfg1 = {"blas": [(verts, np.arange(1, len(verts) + 1, dtype=np.uint32))], "instances": [(1, sc.IDENTITY3x4[None], np.zeros(1, np.uint32))]}
    t1, o1 = build_product(rc, cfg1), build_oracle(oracle, cfg1)
    t1.set_option("pool", pool)
    for k in (-1, 3):
        t1.set_option("kernel", k)
        assert np.array_equal(rc.view_factors(t1, rays_per_triangle=5, seed=3), o1.view_factors(5, seed=3)), k
    b4 = rc.build_blas4(verts, device=0)
    b4._scene.set_option("pool", pool)
    rays = random_rays(rc, 200, 7, [-1, -1, -1], [1, 1, 1])
    o4 = oracle.Scene()
    b = o4.add_blas(verts)
    o4.add_instance(b)
    o4.build()
    assert_hits_equal(b4.trace(rays), o4.trace4(b, rays), f"bvh4 pool {pool}")
    assert t1.get_option("claim_drift") == 0
    t1.free()


def test_claim_counters_are_never_reset_and_never_drift(rc, oracle):
    """The chunk counters count on from launch to launch (no memset between launches): after hundreds of launches of every
    persistent kernel, driver and batch size -- more than the 64 counter slots, so every slot is reused several times -- the
    device counters equal the host's image of them, and results stay bit-exact."""
    import torch
    cfg = rc.scenes.config_c3(lattice=(3, 3, 2))
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    rays = rc.scenes.c3_primary_rays(cfg, 160, 120)
    want, want_any = o.trace(rays, nthreads=8), o.trace(rays, mode="any", nthreads=8)
    g = np.random.default_rng(9)
    for it in range(150):
        k = int(g.choice([-1, 0, 1, 2, 3, 4, 5, 6]))
        n = int(g.choice([1, 7, 64, 129, 1000, len(rays)]))
        t.set_option("kernel", k)
        if it % 3 == 0:
            assert_hits_equal(t.trace(rays[:n], mode="any"), want_any[:n], f"iteration {it} any kernel {k} n {n}")
        else:
            assert_hits_equal(t.trace(rays[:n]), want[:n], f"iteration {it} kernel {k} n {n}")
        if it % 25 == 24:
            assert t.get_option("claim_drift") == 0, it
    t.set_option("kernel", -1)
    il = rc.get_illumination(t, [0.1, 0.2, 1.0], 64)
    assert np.array_equal(il, o.get_illumination([0.1, 0.2, 1.0], 64, nthreads=8))
    assert t.get_option("claim_drift") == 0

    # several streams: slots are shared by launches 64 apart, whatever stream they ran on
    streams = [torch.cuda.Stream() for _ in range(3)]
    dr = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
    outs = [torch.empty(len(rays) * 32, dtype=torch.uint8, device="cuda") for _ in range(3)]
    torch.cuda.synchronize()
    for it in range(200):
        j = it % 3
        t.set_option("kernel", [5, 3, 1][it % 3])
        t.trace_device(dr.data_ptr(), outs[j].data_ptr(), len(rays), stream=streams[j].cuda_stream)
    torch.cuda.synchronize()
    for j in range(3):
        assert_hits_equal(outs[j].cpu().numpy().view(rc.HIT_DT), want, f"stream {j}")
    assert t.get_option("claim_drift") == 0
    t.wait_for_gpu()
    t.free()


def test_status_word_is_not_cleared_by_later_launches(rc, oracle):
    """The scene's stack-overflow word is sticky: kernels only set it, check_status / rc_wait clear it when they report it.  (No
    LBVH over 30-bit Morton codes + indices is deeper than the 128 stack entries, so a real overflow cannot be built; the word is
    set by hand and must survive further launches until rc_wait reports it once.)"""
    import ctypes
    import torch
    sc = rc.scenes
    t = rc.TLAS(0)
    t.push(np.array([[0, 0, 0, 1, 0, 0, 0, 1, 0]], np.float32), sc.IDENTITY3x4[None])
    t.sync()
    rays = sc.make_rays(np.array([[0.2, 0.2, -1.0]] * 300), np.array([[0.0, 0.0, 1.0]] * 300))
    dr = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
    dh = torch.empty(len(rays) * 32, dtype=torch.uint8, device="cuda")
    t.trace_device(dr.data_ptr(), dh.data_ptr(), len(rays))
    t.wait_for_gpu()  # nothing pending: no error
    assert dh.cpu().numpy().view(rc.HIT_DT)["hit"].all()
    t.set_option("debug_set_overflow", 1)
    for k in (0, 1, 3, 5) * 20:  # 80 asynchronous launches, more than the 64 counter slots: none of them may clear the report
        t.set_option("kernel", k)
        t.trace_device(dr.data_ptr(), dh.data_ptr(), len(rays))
    with pytest.raises(rc.RaycoreError, match="overflow"):
        t.wait_for_gpu()
    t.wait_for_gpu()  # reported once, then clear
    t.set_option("debug_set_overflow", 1)
    with pytest.raises(rc.RaycoreError, match="overflow"):
        t.trace(rays)  # the synchronous entry points look at the same word
    assert t.trace(rays)["hit"].all()
    t.free()
