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

    # rows_sharded on one rank: sources are addressed in metadata order, so block row r IS matrix row r (metadata r + 1)
    block, row_index = rd.view_factors_distributed(t, rpt, seed, mode="rows_sharded")
    torch.cuda.synchronize()
    assert block.shape == (n, n) and np.array_equal(row_index, np.arange(n))
    g = np.random.default_rng(5)
    for src in g.choice(n, 64, replace=False):   # flat (Morton-sorted) primitive index, the oracle's addressing
        want = o.view_factor_row(rpt, int(src), seed=seed)
        got = block[int(meta[src] - 1)].cpu().numpy().view(np.uint32)
        assert np.array_equal(got, want), f"source primitive {src}: {int((got != want).sum())} entries differ"
        assert got[meta[src] - 1] == 0 and got.sum() <= rpt
    row_sums = block.sum(dim=1, dtype=torch.int64)
    assert int(row_sums.max()) <= rpt and int(row_sums.min()) >= 0
    total = int(row_sums.sum())
    assert total > 0.5 * n * rpt  # a closed room: most rays hit something that is not their source
    assert int(torch.diagonal(block).abs().sum()) == 0   # hit_meta != src_meta (:94)
    # a checksum per row to compare the other partitions without holding two 10 GB matrices
    w = torch.arange(1, n + 1, device=block.device, dtype=torch.int64)
    want_sum = row_sums.cpu().numpy()
    want_chk = torch.cat([(block[a:a + 4096].to(torch.int64) * w).sum(dim=1) for a in range(0, n, 4096)]).cpu().numpy()
    del block, row_sums
    torch.cuda.empty_cache()
    for mode in ("rows", "rays"):
        m = rd.view_factors_distributed(t, rpt, seed, mode=mode)
        torch.cuda.synchronize()
        assert m.shape == (n, n)
        assert np.array_equal(m.sum(dim=1, dtype=torch.int64).cpu().numpy(), want_sum), mode
        chk = torch.cat([(m[a:a + 4096].to(torch.int64) * w).sum(dim=1) for a in range(0, n, 4096)]).cpu().numpy()  # in row slabs: no 20 GB temporaries
        assert np.array_equal(chk, want_chk), mode
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


def test_status_word_is_not_cleared_by_later_launches(rc, oracle, monkeypatch):
    """The scene's stack-overflow word is sticky: kernels only set it, check_status / rc_wait clear it when they report it.  (No
    LBVH over 30-bit Morton codes + indices is deeper than the 128 stack entries, so a real overflow cannot be built; the word is
    set by hand -- a test hook the library only honours under RC_ENABLE_DEBUG_HOOKS=1 -- and must survive further launches until
    rc_wait reports it once.)"""
    import ctypes
    import torch
    sc = rc.scenes
    t = rc.TLAS(0)
    monkeypatch.delenv("RC_ENABLE_DEBUG_HOOKS", raising=False)
    with pytest.raises(rc.RaycoreError, match="unknown option"):
        t.set_option("debug_set_overflow", 1)  # not part of the product's interface
    monkeypatch.setenv("RC_ENABLE_DEBUG_HOOKS", "1")
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


def test_generic_triangle_metadata(rc, oracle):
    """Triangle{TMetadata} for a metadata type other than UInt32 (src/triangle_mesh.jl:1-7, TLAS(items, metadata_fn) :2276-2324 takes
    TMetadata = typeof(metadata_fn(1, 1))): the library stores a uint32 word per primitive, the host mirror interns the values."""
    sc = rc.scenes
    items = [sc.fan_sphere(8, 5, centre=(0, 0, 0), radius=0.5), sc.fan_sphere(8, 5, centre=(2, 0, 0), radius=0.5)]
    acc = rc.TLAS_from_items(items, lambda mi, fi: ("mesh%d" % mi, fi * 0.5))
    t = acc._owner
    assert t.eltype() == ("Triangle", tuple)
    hit, tri, dist, bary, inst = rc.closest_hit(acc, rc.Ray((2.0, 0.05, -3.0), (0.0, 0.0, 1.0)))
    assert hit and inst == 2 and tri.metadata[0] == "mesh2" and isinstance(tri.metadata[1], float)
    # the same scene with integer metadata: same hit, the word is the value itself
    acc_u = rc.TLAS_from_items(items, lambda mi, fi: 1000 * mi + fi)
    hit_u, tri_u, dist_u, _, inst_u = rc.closest_hit(acc_u, rc.Ray((2.0, 0.05, -3.0), (0.0, 0.0, 1.0)))
    assert hit_u and inst_u == 2 and dist_u == dist and tri_u.metadata // 1000 == 2
    assert tri.metadata[1] == (tri_u.metadata % 1000) * 0.5          # the same face, through the table
    assert acc_u._owner.eltype() == ("Triangle", np.uint32)
    miss = rc.closest_hit(acc, rc.Ray((9.0, 9.0, -3.0), (0.0, 0.0, 1.0)))
    assert not miss[0] and miss[4] == 0
    grid = rc.hits_from_grid(acc, (0.0, 0.0, 1.0), grid_size=16)   # one return type whatever the metadata type
    assert isinstance(grid, np.ndarray) and grid.dtype == rc.RAYHIT_DT
    typed = rc.typed_hit_metadata(acc, grid)
    assert grid["hit"].any() and all((typed[i] is not None) == bool(grid["hit"][i]) for i in np.ndindex(grid.shape))
    pts, centre = rc.get_centroid(acc, (0.0, 0.0, 1.0), grid_size=16)
    assert len(pts) == int(grid["hit"].sum()) and np.all(np.isfinite(centre))


def test_view_factor_source_addressing_and_general_metadata(rc, oracle):
    """RC_VF_SOURCES_BY_METADATA (a contiguous source range = a contiguous block of final rows when the metadata are a permutation
    of 1..N) against the plain matrix, with shuffled metadata; and metadata with duplicates / out-of-range ids through the
    multi-GPU driver's general path (src/kernels.jl:85-97: result[src_meta, hit_meta], several faces may share a row)."""
    import torch
    from raycore_jl_amd import distributed as rd
    from raycore_jl_amd._capi import check, lib, ptr
    sc = rc.scenes
    verts = np.concatenate([sc.fan_sphere(10, 6, radius=0.5), sc.box_room((-1.5, -1.5, -1.5), (1.5, 1.5, 1.5), 2)])
    n = len(verts)
    perm_meta = (np.random.default_rng(3).permutation(n) + 1).astype(np.uint32)
    cfg = {"blas": [(verts, perm_meta)], "instances": [(1, sc.IDENTITY3x4[None], np.zeros(1, np.uint32))]}
    t = build_product(rc, cfg)
    want = rc.view_factors(t, rays_per_triangle=64, seed=5)                    # [src_meta-1, hit_meta-1]
    m = torch.zeros(n * n, dtype=torch.int32, device="cuda")
    a = n // 3
    for s0, s1 in ((0, a), (a, n)):                                            # two row blocks, each written at its own offset
        check(lib().rc_view_factors_device(t._h, 64, 5, s0, s1, 0, 64, ptr(m[s0 * n:].data_ptr()), n, 1, s0, 2, None))
    torch.cuda.synchronize()
    assert np.array_equal(m.cpu().numpy().view(np.uint32).reshape(n, n), want)
    for mode in ("rows", "rays"):
        out = rd.view_factors_distributed(t, 64, 5, mode=mode, chunks=7)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint32), want), mode
    t.free()
    dup_meta = ((np.arange(n) // 3) + 1).astype(np.uint32)
    dup_meta[5] = n + 100                                                      # out of range: dropped, as an out-of-bounds index would be
    cfg = {"blas": [(verts, dup_meta)], "instances": [(1, sc.IDENTITY3x4[None], np.zeros(1, np.uint32))]}
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    want = o.view_factors(32, seed=8, nthreads=8)
    assert np.array_equal(rc.view_factors(t, rays_per_triangle=32, seed=8), want)
    for mode in ("rows", "rays"):
        out = rd.view_factors_distributed(t, 32, 8, mode=mode)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint32), want), mode
    t.free()


def test_pinned_host_buffers(rc, oracle):
    """rc_host_register / rc_host_unregister: a page-locked ray / hit array pair goes through the same host-buffer entry points with the
    same results (single-launch path and the chunked three-stage pipeline), registering twice is an error, and after unregistering the
    arrays are ordinary pageable memory again."""
    cfg = rc.scenes.config_c3(lattice=(3, 3, 2))
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    for w, h in ((256, 256), (2048, 1600)):     # 65 k rays: one launch; 3.3 M rays: the pipelined path (>= 3 Mi rays)
        rays = rc.scenes.c3_primary_rays(cfg, w, h)
        out = np.empty(len(rays), dtype=rc.HIT_DT)
        want = t.trace(rays).copy()
        if len(rays) < 100_000:
            assert_hits_equal(want, o.trace(rays, nthreads=8), "pageable vs oracle")
        t.host_register(rays); t.host_register(out)
        try:
            got = t.trace(rays, out=out)
            assert got is out and got.tobytes() == want.tobytes()
            assert t.trace(rays, mode="any", out=out)["hit"].tobytes() == t.trace(rays, mode="any")["hit"].tobytes()
            with pytest.raises(rc.RaycoreError):
                t.host_register(rays)
        finally:
            t.host_unregister(out); t.host_unregister(rays)
        with pytest.raises(rc.RaycoreError):
            t.host_unregister(rays)
        assert t.trace(rays).tobytes() == want.tobytes()
    t.free()


def test_option_clamps(rc):
    """rc_set_option validates what it stores: blocks_per_cu is bounded by what the stack spill area is sized for, claim
    sizes below 16 rays are raised, the kernel selector stays in range, claim_shards is a power of two <= 16."""
    t = rc.TLAS(0)
    for name, given, stored in (("blocks_per_cu", 100, 8), ("blocks_per_cu", -3, 0), ("pool", 3, 16), ("pool", 0, 0), ("pool", 1 << 30, 1 << 20),
                                ("kernel", 99, 6), ("kernel", -7, -1), ("claim_shards", 5, 4), ("claim_shards", 1000, 16), ("refill", 0, 1), ("refill", 999, 64),
                                ("sched_thr", 0, 1), ("sched_thr", 65, 64)):
        t.set_option(name, given)
        assert t.get_option(name) == stored, (name, given, t.get_option(name))
    with pytest.raises(rc.RaycoreError):
        t.set_option("no_such_option", 1)
    t.free()


def test_drivers_against_closed_forms(rc):
    """The HIP drivers against answers that come from neither implementation (tests/test_oracle_analytic_drivers.py has the
    derivations): view_factors' uniform-hemisphere sampling => solid angle of a square over 2 pi; get_illumination => the grid points
    inside a rectangle facing the view direction."""
    a, h, eps, rays = 1.0, 1.0, 1e-3, 400_000
    src = np.array([[-eps, -eps, 0, eps, -eps, 0, 0, eps, 0]], np.float32)
    p = [(-a, -a, h), (a, -a, h), (a, a, h), (-a, a, h)]
    target = np.array([(p[0], p[2], p[1]), (p[0], p[3], p[2])], np.float32).reshape(-1, 9)
    t = rc.TLAS(0)
    t.add_geometry(np.concatenate([src, target]), np.array([1, 2, 3], np.uint32))
    t.push_instances(1)
    t.sync()
    row = rc.view_factors(t, rays, seed=11)[0].astype(np.int64)
    want = 4.0 * np.arctan(a * a / ((h - 0.01) * np.sqrt(2 * a * a + (h - 0.01) ** 2))) / (2 * np.pi)
    assert row[0] == 0 and abs(row[1:].sum() / rays - want) < 5 * np.sqrt(want * (1 - want) / rays)
    t.free()
    grid = 220
    t = rc.TLAS(0)
    t.add_geometry(np.array([[0, 0, 0, 1, 0, 0, 1, 2, 0], [0, 0, 0, 1, 2, 0, 0, 2, 0]], np.float32), np.array([1, 2], np.uint32))
    t.push_instances(1)
    t.sync()
    counts = rc.get_illumination(t, (0, 0, -1), grid)
    n_long = sum(1 for i in range(1, grid + 1) if abs((i - (grid + 1) / 2) * (2.2 / grid)) <= 1.0)
    n_short = sum(1 for i in range(1, grid + 1) if abs((i - (grid + 1) / 2) * (1.2 / grid)) <= 0.5)
    assert abs(int(counts.sum()) - n_long * n_short) <= max(n_long, n_short)
    pts, centre = rc.get_centroid(t, (0, 0, -1), grid_size=64)  # hit points of a symmetric grid over the rectangle: their mean is its centre
    assert len(pts) > 1000 and np.allclose(centre, (0.5, 1.0, 0.0), atol=2e-2) and np.abs(pts[:, 2]).max() < 1e-5
    t.free()


def test_timeline_instrumentation_keeps_results(rc):
    """Option "timeline_ptr" (dev, tools/archive/timeline_probe.py): kernel 5 built with per-wave event stamps writes 8 words per wave into the
    caller's buffer and returns the same hits."""
    import torch
    cfg = rc.scenes.config_c3(lattice=(3, 3, 2))
    t = build_product(rc, cfg)
    rays = rc.scenes.c3_primary_rays(cfg, 512, 512)
    n = len(rays)
    dr = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
    a, b = (torch.empty(n * 32, dtype=torch.uint8, device="cuda") for _ in range(2))
    t.set_option("kernel", 5)
    t.trace_device(dr.data_ptr(), a.data_ptr(), n)
    waves = t.get_option("n_cus") * 24
    buf = torch.zeros(waves * 8, dtype=torch.int64, device="cuda")
    t.set_option("timeline_ptr", buf.data_ptr())
    t.trace_device(dr.data_ptr(), b.data_ptr(), n)
    t.set_option("timeline_ptr", 0)
    torch.cuda.synchronize()
    assert a.cpu().numpy().tobytes() == b.cpu().numpy().tobytes()
    w = buf.cpu().numpy().reshape(-1, 8)
    ran = w[w[:, 4] != 0]
    assert len(ran) == min(waves, (n + 767) // 768 * 12)
    assert (ran[:, 4] >= ran[:, 0]).all() and ((ran[:, 1] == 0) | (ran[:, 1] >= ran[:, 0])).all()  # end after start, dry time after start
    t.free()


def test_device_entry_points_reject_null_buffers(rc):
    """The device-pointer entry points are asynchronous: a NULL ray / hit / output pointer must be refused on the host (an error code
    and a message), not handed to a kernel."""
    from raycore_jl_amd._capi import lib
    import ctypes as C
    t = build_product(rc, rc.scenes.config_c3(lattice=(2, 2, 1)))
    L = lib()
    for fn in (L.rc_trace_closest_device, L.rc_trace_any_device):
        assert fn(t._h, None, None, 64, None) == 1  # RC_ERR_INVALID_ARGUMENT
        assert fn(t._h, None, None, 0, None) == 0  # an empty batch needs no buffers
    view = (C.c_float * 3)(0.0, 0.0, 1.0)
    assert L.rc_generate_ray_grid_device(t._h, view, 8, None, None) != 0
    assert L.rc_get_illumination_device(t._h, view, 8, 0, 64, None, None) != 0
    assert b"NULL" in L.rc_last_error()
    t.free()


def test_trace_launches_are_hipgraph_capturable(rc, oracle):
    """A frame of primary trace -> shadow-ray generation -> any_hit, captured into a hipGraph and replayed: the claim counters reset
    themselves inside the kernels and a launch leaves no host-side state behind, so a replay is as good as a fresh launch."""
    import torch
    sc = rc.scenes
    cfg = sc.config_c3(lattice=(3, 3, 2))
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    rays = sc.c3_primary_rays(cfg, 320, 200)
    n = len(rays)
    want = o.trace(rays, nthreads=8)
    light = np.array([10, 10, 10], np.float32)
    want_occ = o.trace(o.shadow_rays(rays, want, light, 1e-3), mode="any", nthreads=8)
    dr = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
    dh, dsh, docc = (torch.empty(n * 32, dtype=torch.uint8, device="cuda") for _ in range(3))
    s = torch.cuda.Stream()

    def frame(stream):
        t.trace_device(dr.data_ptr(), dh.data_ptr(), n, stream=stream)
        t.shadow_rays_device(dr.data_ptr(), dh.data_ptr(), n, light, dsh.data_ptr(), bias=1e-3, stream=stream)
        t.trace_device(dsh.data_ptr(), docc.data_ptr(), n, mode="any", stream=stream)

    with torch.cuda.stream(s):
        frame(s.cuda_stream)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    bystander = build_product(rc, sc.config_c3(lattice=(2, 1, 1)))
    with torch.cuda.graph(g, stream=s):
        frame(torch.cuda.current_stream().cuda_stream)
        # entry points that allocate, copy and free run beside an open (global-mode) capture without invalidating it: they switch the
        # calling thread's capture interaction mode to relaxed (a finaliser destroying a scene mid-capture is the everyday case)
        bystander.trace(rays[:1000])
        big_batch = np.concatenate([rays] * 52)[:3_300_000]       # >= 3 Mi rays: the pipelined path with its upload / download threads
        assert len(big_batch) == 3_300_000
        got_big = bystander.trace(big_batch)
        twin = build_product(rc, sc.config_c3(lattice=(2, 1, 1)))
        vf2 = rc.view_factors_multi([bystander, twin], 8, seed=3)   # one worker thread per scene
        assert np.array_equal(got_big[:len(rays)], got_big[len(rays):2 * len(rays)]) and np.array_equal(vf2, rc.view_factors(twin, 8, seed=3))
        twin.free()
        bystander.free()
        newcomer = build_product(rc, sc.config_c3(lattice=(1, 1, 1)))
        frame(torch.cuda.current_stream().cuda_stream)
    newcomer.free()
    for rep in range(4):
        dh.zero_(); docc.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert_hits_equal(dh.cpu().numpy().view(rc.HIT_DT), want, f"replay {rep} primary")
        assert np.array_equal(docc.cpu().numpy().view(rc.HIT_DT)["hit"], want_occ["hit"]), rep
    assert t.get_option("claim_drift") == 0
    # a graph bakes its counter slots in.  Replays on one stream while more than a full rotation of eager launches (the eager
    # slots: 48) runs on another must never meet on a slot: captured launches have slots of their own.
    big = sc.c3_primary_rays(cfg, 1500, 1000)
    want_big = o.trace(big, nthreads=16)
    dbig = torch.from_numpy(big.view(np.uint8).reshape(-1)).cuda()
    hbig = torch.empty(len(big) * 32, dtype=torch.uint8, device="cuda")
    e = torch.cuda.Stream()
    for rep in range(3):
        dh.zero_(); docc.zero_(); hbig.zero_()
        torch.cuda.synchronize()
        for k in range(60):
            t.trace_device(dbig.data_ptr(), hbig.data_ptr(), len(big), stream=e.cuda_stream)
            if k % 6 == 0:
                g.replay()
        torch.cuda.synchronize()
        assert_hits_equal(dh.cpu().numpy().view(rc.HIT_DT), want, f"replay beside eager launches {rep}")
        assert np.array_equal(docc.cpu().numpy().view(rc.HIT_DT)["hit"], want_occ["hit"])
        assert_hits_equal(hbig.cpu().numpy().view(rc.HIT_DT), want_big, f"eager launches beside replays {rep}")
    assert t.get_option("claim_drift") == 0
    # a capture on a stream the scene has never launched on works too: every captured launch gets a stack spill region of its own,
    # allocated inside the capture (the entry points run in relaxed capture-interaction mode)
    fresh = torch.cuda.Stream()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=fresh):
        t.trace_device(dr.data_ptr(), dh.data_ptr(), n, stream=torch.cuda.current_stream().cuda_stream)
    dh.zero_()
    g2.replay()
    torch.cuda.synchronize()
    assert_hits_equal(dh.cpu().numpy().view(rc.HIT_DT), want, "graph captured on a fresh stream")
    # the pool: 16 captured launches per scene (2 x 3 in g, 1 in g2 so far); the 17th is refused with an explanation, and
    # "release_captures" (the caller's promise that its graphs are gone) hands regions and slots out again
    held = t.get_option("release_captures")
    assert held == 5   # two frames x (closest + any) in g, one in g2
    graphs = []
    with pytest.raises(rc.RaycoreError, match="captured launches"):
        for _ in range(14):
            gk = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gk, stream=fresh):
                t.trace_device(dr.data_ptr(), dh.data_ptr(), n, stream=torch.cuda.current_stream().cuda_stream)
            graphs.append(gk)
    assert len(graphs) == 16 - held and t.get_option("release_captures") == 16
    torch.cuda.synchronize()
    del graphs, g, g2, gk
    t.set_option("release_captures", 1)
    assert t.get_option("release_captures") == 0
    g3 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g3, stream=fresh):
        t.trace_device(dr.data_ptr(), dh.data_ptr(), n, stream=torch.cuda.current_stream().cuda_stream)
    dh.zero_()
    g3.replay()
    torch.cuda.synchronize()
    assert_hits_equal(dh.cpu().numpy().view(rc.HIT_DT), want, "graph captured after release_captures")
    # a captured launch's spill region is sized for ITS grid (every capture used to pin the largest grid's 268 MB), and launches
    # are handed back one by one with the token the scene gave right after the capture -- other graphs stay alive
    tok3, bytes3 = t.get_option("last_capture_token"), t.get_option("capture_bytes")
    assert tok3 >= 1 and t.get_option("release_captures") == 1 and bytes3 > 0
    small = 4096
    g4 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g4, stream=fresh):
        t.trace_device(dr.data_ptr(), dh.data_ptr(), small, stream=torch.cuda.current_stream().cuda_stream)
    tok4 = t.get_option("last_capture_token")
    assert tok4 not in (0, tok3) and t.get_option("release_captures") == 2
    assert t.get_option("capture_bytes") - bytes3 < bytes3 // 8          # 4096 rays: a few blocks' worth of spill area, not the whole machine's
    del g3
    t.set_option("release_capture", tok3)
    assert t.get_option("release_captures") == 1 and t.get_option("last_capture_token") == tok4
    with pytest.raises(rc.RaycoreError, match="token"):
        t.set_option("release_capture", tok3)                            # already handed back
    dh.zero_()
    g4.replay()                                                          # the other graph is untouched
    torch.cuda.synchronize()
    assert_hits_equal(dh.cpu().numpy().view(rc.HIT_DT)[:small], want[:small], "graph that stayed alive beside a released one")
    assert t.get_option("claim_drift") == 0
    t.free()


def test_graphs_captured_on_one_stream_replay_concurrently_on_deep_trees(rc, oracle):
    """graphs captured on the SAME stream used to share that stream's stack spill region; replayed side by side on a tree
    deeper than the LDS lane stack they overwrote each other's entries (silently wrong hits).  Every captured launch now owns its
    region: two graphs (different ray sets) captured on one stream, replayed at once on two streams beside eager launches on a third,
    each equal to the oracle."""
    import torch

    def chain(levels, fat=0.3):
        tris = []
        for j in range(1, levels + 1):
            for axis in range(3):
                size = 2.0 ** (-j + 1)
                p = np.full(3, fat * size); p[axis] = size
                q = p.copy(); q[(axis + 1) % 3] += 0.5 * size * fat
                r = np.full(3, -1e-4 * (1 + 0.5 * j))
                tris.append(np.concatenate([r, p, q]))
        return np.array(tris, dtype=np.float32)
    xf = np.tile(rc.scenes.IDENTITY3x4, (4, 1)).astype(np.float32)
    xf[1, [0, 5, 10]] = 0.5
    xf[2, [0, 5, 10]] = 0.25
    xf[3, [3, 7, 11]] = [0.01, 0.0, 0.0]
    cfg = {"blas": [(chain(10), None)], "instances": [(1, xf, np.arange(4, dtype=np.uint32))]}
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    g = rc.scenes.rng(77)
    n = 400_000   # long enough launches that the two replays overlap
    sets = []
    for k in range(3):
        org = g.uniform(-0.2, 0.0, size=(n, 3))
        d = rc.scenes.normalize(g.uniform(0.05, 1.0, size=(n, 3)))
        rays = rc.scenes.make_rays(org, d)
        sets.append((rays, o.trace(rays, nthreads=16), torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda(),
                     torch.empty(n * 32, dtype=torch.uint8, device="cuda")))
    cap, s1, s2, s3 = (torch.cuda.Stream() for _ in range(4))
    graphs = []
    for k in range(2):
        gk = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gk, stream=cap):   # both on the SAME capture stream
            for _ in range(3):
                t.trace_device(sets[k][2].data_ptr(), sets[k][3].data_ptr(), n, stream=torch.cuda.current_stream().cuda_stream)
        graphs.append(gk)
    for rep in range(5):
        for k in range(3):
            sets[k][3].zero_()
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            graphs[0].replay()
        with torch.cuda.stream(s2):
            graphs[1].replay()
        for _ in range(3):
            t.trace_device(sets[2][2].data_ptr(), sets[2][3].data_ptr(), n, stream=s3.cuda_stream)
        torch.cuda.synchronize()
        for k in range(3):
            assert_hits_equal(sets[k][3].cpu().numpy().view(rc.HIT_DT), sets[k][1], f"rep {rep} set {k}")
    assert t.get_option("claim_drift") == 0
    del graphs
    t.free()


def test_cost_ordered_claiming_and_tapered_chunks_change_no_result(rc, oracle):
    """Scheduling knobs of the persistent kernels: guided chunk sizes (option "taper") and cost-ordered claiming (option "cost_order": the
    chunks that held long rays in the previous launch of the same shape are claimed first, through a permutation built on the device).
    Per-ray results cannot depend on either: repeated launches of one batch -- the second and later ones run in the learned order --
    equal the oracle bit for bit, for closest and any hit, the three phased kernels, two batch sizes interleaved on two streams, and the
    claim counters end at zero (a claim order that is not a permutation would skip or repeat chunks)."""
    import torch
    cfg = rc.scenes.config_c2(30_000, 100)
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    wb = t.world_bound()
    batches = [random_rays(rc, 900_000, 11, wb.p_min, wb.p_max), random_rays(rc, 1_300_000, 12, wb.p_min, wb.p_max)]
    want = [(o.trace(b, nthreads=16), o.trace(b, mode="any", nthreads=16)) for b in batches]
    d_rays = [torch.from_numpy(b.view(np.uint8).reshape(-1)).cuda() for b in batches]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [torch.empty(len(b) * 32, dtype=torch.uint8, device="cuda") for b in batches]
    for kernel, taper, cost_order in ((-1, 12, 1), (3, 12, 1), (5, 0, 1), (5, 32, 1), (-1, 12, 0), (-1, 5, 1)):
        t.set_option("kernel", kernel); t.set_option("taper", taper); t.set_option("cost_order", cost_order)
        for rep in range(4):
            for mode_i, mode in enumerate(("closest", "any")):
                for j in (0, 1):
                    outs[j].zero_()
                torch.cuda.synchronize()
                for j in (0, 1):
                    t.trace_device(d_rays[j].data_ptr(), outs[j].data_ptr(), len(batches[j]), mode=mode, stream=streams[j].cuda_stream)
                torch.cuda.synchronize()
                for j in (0, 1):
                    got = outs[j].cpu().numpy().view(rc.HIT_DT)
                    if mode == "closest":
                        assert_hits_equal(got, want[j][0], f"kernel {kernel} taper {taper} cost_order {cost_order} rep {rep} batch {j}")
                    else:
                        assert np.array_equal(got["hit"], want[j][1]["hit"]), f"any: kernel {kernel} taper {taper} cost_order {cost_order} rep {rep} batch {j}"
        assert t.get_option("claim_drift") == 0
    # the claim order the device built for the last shape is a permutation of its chunks (read back through the dev options)
    import ctypes
    n_order, p_order = t.get_option("debug_order_n"), t.get_option("debug_order_ptr")
    assert n_order == -(-len(batches[1]) // 128) and p_order
    order = torch.empty(n_order, dtype=torch.int32, device="cuda")
    ctypes.CDLL("libamdhip64.so").hipMemcpy(ctypes.c_void_p(order.data_ptr()), ctypes.c_void_p(p_order), ctypes.c_size_t(4 * n_order), 3)
    torch.cuda.synchronize()
    order = order.cpu().numpy()
    assert np.array_equal(np.sort(order), np.arange(n_order)) and not np.array_equal(order, np.arange(n_order))  # ... and not the identity: something was learned
    # more launch shapes than history entries (8): the least recently used one is dropped, nothing breaks
    t.set_option("kernel", -1); t.set_option("taper", 12); t.set_option("cost_order", 1)
    for k in range(12):
        n = 700_000 + 10_007 * k
        for rep in range(2):
            t.trace_device(d_rays[1].data_ptr(), outs[1].data_ptr(), n, stream=streams[0].cuda_stream)
        torch.cuda.synchronize()
        assert_hits_equal(outs[1].cpu().numpy().view(rc.HIT_DT)[:n], want[1][0][:n], f"shape {k}")
    assert t.get_option("claim_drift") == 0
    t.free()


def test_changing_batch_sizes_never_block_and_change_no_result(rc, oracle):
    """cost-ordered claiming keeps 8 histories keyed by launch shape.  A workload whose batch size changes with every launch (a
    wavefront tracer compacting its rays bounce by bounce) used to evict -- stream synchronise, 3 x hipFree, 3 x hipMalloc -- on every
    launch after the eighth.  Now a miss never blocks: one-off shapes run in natural order, a shape that comes back takes over the entry
    of an idle or same-stream shape with no allocation.  Results are the oracle's whatever order the chunks were claimed in."""
    import time
    import torch
    cfg = rc.scenes.config_c3(lattice=(4, 4, 2))
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    rays = rc.scenes.c3_primary_rays(cfg, 1024, 600)
    want = o.trace(rays, nthreads=16)
    dr = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
    s = torch.cuda.Stream()
    sizes = [len(rays) - 4096 * k for k in range(24)]           # 24 distinct chunk counts on one stream: three times the table
    outs = [torch.empty(n * 32, dtype=torch.uint8, device="cuda") for n in sizes]
    t.trace_device(dr.data_ptr(), outs[0].data_ptr(), sizes[0], stream=s.cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stamps = []
    for n, out in zip(sizes, outs):
        t.trace_device(dr.data_ptr(), out.data_ptr(), n, stream=s.cuda_stream)
        stamps.append(time.perf_counter())
    enqueue_s = stamps[-1] - t0
    torch.cuda.synchronize()
    total_s = time.perf_counter() - t0
    for n, out in zip(sizes, outs):
        assert_hits_equal(out.cpu().numpy().view(rc.HIT_DT), want[:n], f"n = {n}")
    # the first eight shapes allocate their history buffers (hipMalloc, no synchronisation); from the ninth on a launch only enqueues: a
    # stream synchronise or an eviction that waited would cost at least the ~3 ms of device work queued in front of it
    late_s = stamps[-1] - stamps[7]
    assert late_s < 1.5e-3, f"enqueueing launches 9-24 took {late_s * 1e3:.2f} ms (all 24: {enqueue_s * 1e3:.2f} of {total_s * 1e3:.2f} ms): the enqueue waited for the device"
    # two shapes that keep coming back after the table is full of one-offs: they are recognised and learn (third launch runs ordered)
    a, b = sizes[20], sizes[22]
    for rep in range(4):
        for n in (a, b):
            out = outs[sizes.index(n)]
            out.zero_()
            t.trace_device(dr.data_ptr(), out.data_ptr(), n, stream=s.cuda_stream)
    torch.cuda.synchronize()
    for n in (a, b):
        assert_hits_equal(outs[sizes.index(n)].cpu().numpy().view(rc.HIT_DT), want[:n], f"recurring n = {n}")
    assert t.get_option("claim_drift") == 0
    t.free()


def test_alternating_batches_keep_their_own_claim_order(rc, oracle):
    """VERDICT r3 #5a: consecutive launches of ONE shape (size, mode, stream) that trace DIFFERENT rays -- two cameras, N light samples.  The
    batch is recognised on the device by 64 sample rays (order_select, inside the launch): A and B get a slot each and keep alternating between them, a
    camera that moves a little stays in its slot, a fifth distinct batch evicts the least recently used of the four slots -- and whatever
    order the chunks are claimed in, every launch returns the oracle's hits."""
    import ctypes
    import torch
    sc = rc.scenes
    cfg = sc.config_c3(lattice=(4, 4, 2))
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    centre = cfg["lattice_centre"]

    def view(eye):
        return sc.pinhole_rays(1280, 800, np.asarray(eye, dtype=np.float64), centre, 45.0)   # 1.02 M rays: more chunks than waves, so the claim order applies
    eyes = [centre + np.array([0.0, 0.0, -12.0]), centre + np.array([11.0, 2.0, -4.0]), centre + np.array([-3.0, 10.0, 5.0]),
            centre + np.array([0.5, -12.0, 0.5]), centre + np.array([-11.0, -1.0, 2.0])]
    batches = [view(e) for e in eyes]
    n = len(batches[0])
    want = [o.trace(b, nthreads=16) for b in batches]
    dev = [torch.from_numpy(b.view(np.uint8).reshape(-1)).cuda() for b in batches]
    out = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
    hip = ctypes.CDLL("libamdhip64.so")

    def header():
        h = torch.empty(40, dtype=torch.int32, device="cuda")
        hip.hipMemcpy(ctypes.c_void_p(h.data_ptr()), ctypes.c_void_p(t.get_option("debug_ctl_ptr")), ctypes.c_size_t(160), 3)
        torch.cuda.synchronize()
        return h.cpu().numpy().view(np.uint32)

    def launch(k, what):
        out.zero_()
        t.trace_device(dev[k].data_ptr(), out.data_ptr(), n)
        torch.cuda.synchronize()
        assert_hits_equal(out.cpu().numpy().view(rc.HIT_DT), want[k], what)
        h = header()
        return int(h[0]), int(h[1]), int(h[4]), [int(g) for g in h[12:16]]   # slot, order valid, fresh, launches per slot

    seen = {}
    for rep in range(10):                      # A B A B ...: two slots, each counting its own launches; from its third launch on a batch has an order (the second recorded)
        for k in (0, 1):
            sel, valid, fresh, gens = launch(k, f"alternating rep {rep} batch {k}")
            seen.setdefault(k, sel)
            assert sel == seen[k] and fresh == (1 if rep == 0 else 0) and valid == (0 if rep < 2 else 1) and gens[sel] == rep + 1, (rep, k, sel, valid, fresh, gens)
    assert seen[0] != seen[1]
    # the same camera, moved a little every frame: one slot (its samples follow the camera: nobody else's is taken), but never a REPEAT: every
    # frame starts over in that slot -- natural order, nothing recorded (round 5: decided on the device, so it holds however far ahead the host enqueues)
    moving = [view(eyes[0] + np.array([0.02 * f, 0.01 * f, 0.0])) for f in range(1, 7)]
    for f, b in enumerate(moving):
        d = torch.from_numpy(b.view(np.uint8).reshape(-1)).cuda()
        out.zero_()
        t.trace_device(d.data_ptr(), out.data_ptr(), n)
        torch.cuda.synchronize()
        assert_hits_equal(out.cpu().numpy().view(rc.HIT_DT), o.trace(b, nthreads=16), f"moving camera frame {f}")
        h = header()
        assert int(h[0]) == seen[0] and int(h[4]) == 1 and int(h[1]) == 0 and int(h[5]) == 0 and int(h[36]) == f + 1, (f, h[:8])   # (the streak of launches that are not REPEATS grows: eight of them and the host would pause the mechanism)
    sel, valid, fresh, gens = launch(0, "batch 0, close to where the camera stopped")
    sel, valid, fresh, gens = launch(0, "batch 0 again: a repeat, the streak starts over")
    assert int(header()[36]) == 0
    # five distinct batches on four slots: the least recently used slot is given away, its batch is fresh when it comes back
    for k in (2, 3, 4):
        sel, valid, fresh, gens = launch(k, f"new batch {k}")
        assert fresh == 1 and valid == 0
    sel, valid, fresh, gens = launch(1, "batch 1 after its slot was given away")
    assert fresh == 1
    sel, valid, fresh, gens = launch(1, "batch 1 again")
    assert fresh == 0 and valid == 0
    sel, valid, fresh, gens = launch(1, "batch 1, third launch since it came back")
    assert fresh == 0 and valid == 1
    assert t.get_option("claim_drift") == 0
    t.free()


def test_batches_that_never_repeat_stop_paying_for_the_claim_order(rc, oracle):
    """A path tracer's bounce rays are new every launch: six different batches in rotation on the history's four slots match nothing, ever.
    The launches report the run of non-repeats to the host through a pinned word (order_commit); after eight of them the host keeps the
    shape out of the mechanism (the header's launch clock stops advancing) for its next 64 launches.  Results are the oracle's throughout."""
    import ctypes
    import torch
    sc = rc.scenes
    cfg = sc.config_c3(lattice=(4, 4, 2))
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    centre = cfg["lattice_centre"]
    g = np.random.default_rng(3)
    eyes = [centre + 13.0 * v / np.linalg.norm(v) for v in g.normal(size=(6, 3))]
    batches = [sc.pinhole_rays(1280, 800, e, centre, 45.0) for e in eyes]
    n = len(batches[0])
    want = [o.trace(b, nthreads=16) for b in batches]
    dev = [torch.from_numpy(b.view(np.uint8).reshape(-1)).cuda() for b in batches]
    out = torch.empty(n * 32, dtype=torch.uint8, device="cuda")
    hip = ctypes.CDLL("libamdhip64.so")

    def clock():
        h = torch.empty(40, dtype=torch.int32, device="cuda")
        hip.hipMemcpy(ctypes.c_void_p(h.data_ptr()), ctypes.c_void_p(t.get_option("debug_ctl_ptr")), ctypes.c_size_t(160), 3)
        torch.cuda.synchronize()
        h = h.cpu().numpy().view(np.uint32)
        return int(h[3]), int(h[4]), int(h[36])   # launches seen by the mechanism, fresh, run of non-repeats

    clocks = []
    for k in range(14):
        out.zero_()
        t.trace_device(dev[k % 6].data_ptr(), out.data_ptr(), n)
        torch.cuda.synchronize()
        assert_hits_equal(out.cpu().numpy().view(rc.HIT_DT), want[k % 6], f"launch {k + 1}")
        clocks.append(clock())
    assert [c[0] for c in clocks[:8]] == list(range(1, 9)) and all(c[1] == 1 for c in clocks[:8]) and clocks[6][2] == 7 and clocks[7][2] == 0   # (the eighth non-repeat starts the pause and the count over)
    assert all(c[0] == 8 for c in clocks[8:]), clocks                      # launches 9-14: natural order, outside the mechanism
    # after the pause the shape is really tried again.  The 64 skipped launches pass (6 done above), then a batch that REPEATS must
    # get its order back: matched from its second launch on, an order in use from its third -- the streak counter starts over with the pause,
    # so one unmatched probe launch does not send the shape straight back into the pause.
    for k in range(58):
        t.trace_device(dev[k % 6].data_ptr(), out.data_ptr(), n)
    torch.cuda.synchronize()
    assert clock()[0] == 8
    seen = []
    for k in range(5):
        out.zero_()
        t.trace_device(dev[2].data_ptr(), out.data_ptr(), n)
        torch.cuda.synchronize()
        assert_hits_equal(out.cpu().numpy().view(rc.HIT_DT), want[2], f"after the pause, launch {k + 1}")
        h = torch.empty(40, dtype=torch.int32, device="cuda")
        hip.hipMemcpy(ctypes.c_void_p(h.data_ptr()), ctypes.c_void_p(t.get_option("debug_ctl_ptr")), ctypes.c_size_t(160), 3)
        torch.cuda.synchronize()
        w = h.cpu().numpy().view(np.uint32)
        seen.append((int(w[3]), int(w[1]), int(w[4]), int(w[36])))          # clock, order valid, fresh, streak
    assert [c[0] for c in seen] == [9, 10, 11, 12, 13], seen               # every one of them was seen by the mechanism again
    assert seen[0][3] == 1 and all(c[2] == 0 and c[3] == 0 for c in seen[1:]), seen   # the first is unmatched (streak restarted at 1, not 9), the rest match
    assert [c[1] for c in seen[2:]] == [1, 1, 1], seen                       # an order from the batch's third launch on
    # a batch that DOES repeat on another shape is unaffected (histories are per shape)
    m = n - 4096
    for rep in range(4):
        out.zero_()
        t.trace_device(dev[0].data_ptr(), out.data_ptr(), m)
        torch.cuda.synchronize()
        assert_hits_equal(out.cpu().numpy().view(rc.HIT_DT)[:m], want[0][:m], f"repeating shape, launch {rep + 1}")
    assert t.get_option("claim_drift") == 0
    t.free()



def test_a_camera_that_moves_every_frame_stops_paying_for_the_claim_order(rc, oracle):
    """a camera that moves a little every frame is recognised as the batch of the frame before -- and ran 1.5-2 % SLOWER with
    that batch's claim order than with the order switched off (an order learned from similar rays gains less than its recording launches
    cost).  Only a REPEAT (identical sample rays) continues a slot's history: the moving camera's frames start over in their slot -- natural
    order, nothing recorded -- and are counted; after eight of them the shape's launches leave the mechanism for a while.  A still camera
    keeps its order."""
    import ctypes
    import torch
    sc = rc.scenes
    cfg = sc.config_c3(lattice=(4, 4, 2))
    t, o = build_product(rc, cfg), build_oracle(oracle, cfg)
    centre = cfg["lattice_centre"]
    eye0 = centre + np.array([0.0, 0.0, -12.0])
    hip = ctypes.CDLL("libamdhip64.so")

    def header():
        h = torch.empty(40, dtype=torch.int32, device="cuda")
        hip.hipMemcpy(ctypes.c_void_p(h.data_ptr()), ctypes.c_void_p(t.get_option("debug_ctl_ptr")), ctypes.c_size_t(160), 3)
        torch.cuda.synchronize()
        w = h.cpu().numpy().view(np.uint32)
        return int(w[3]), int(w[1]), int(w[4]), int(w[36])   # launch clock, order valid, fresh, streak of non-repeats

    def frame(eye, what):
        rays = sc.pinhole_rays(1280, 800, eye, centre, 45.0)
        d = torch.from_numpy(rays.view(np.uint8).reshape(-1)).cuda()
        out = torch.zeros(len(rays) * 32, dtype=torch.uint8, device="cuda")
        t.trace_device(d.data_ptr(), out.data_ptr(), len(rays))
        torch.cuda.synchronize()
        assert_hits_equal(out.cpu().numpy().view(rc.HIT_DT), o.trace(rays, nthreads=16), what)
        return header()
    seen = [frame(eye0, f"still camera, launch {k + 1}") for k in range(4)]
    assert [c[0] for c in seen] == [1, 2, 3, 4] and [c[3] for c in seen] == [1, 0, 0, 0] and seen[3][1] == 1, seen   # repeats: the order is in use, the streak stays at 0
    seen = [frame(eye0 + np.array([0.02 * f, 0.01 * f, 0.0]), f"moving camera, frame {f}") for f in range(1, 13)]
    assert [c[0] for c in seen[:8]] == list(range(5, 13)) and all(c[2] == 1 and c[1] == 0 for c in seen[:8]), seen   # seen by the mechanism: fresh in their slot, natural order
    assert [c[3] for c in seen[:8]] == list(range(1, 8)) + [0], seen                                                  # ... but not repeats (the eighth starts the pause and the count over)
    assert all(c[0] == 12 for c in seen[8:]), seen                                                                    # frames 9-12: outside the mechanism
    assert t.get_option("claim_drift") == 0
    t.free()
    # the same for a caller that never waits: 4 still launches and 20 moving frames enqueued back to back.  The pause is counted and started
    # on the device (order_commit), so it begins after the eighth moving frame however late the host hears of it
    t = build_product(rc, cfg)
    still = sc.pinhole_rays(1280, 800, eye0, centre, 45.0)
    bufs = [torch.from_numpy(sc.pinhole_rays(1280, 800, eye0 + np.array([0.02 * f, 0.01 * f, 0.0]), centre, 45.0).view(np.uint8).reshape(-1)).cuda() for f in range(1, 21)]
    d_still = torch.from_numpy(still.view(np.uint8).reshape(-1)).cuda()
    out = torch.zeros(len(still) * 32, dtype=torch.uint8, device="cuda")
    for _ in range(4):
        t.trace_device(d_still.data_ptr(), out.data_ptr(), len(still))
    for b in bufs:
        t.trace_device(b.data_ptr(), out.data_ptr(), len(still))
    torch.cuda.synchronize()
    h = torch.empty(40, dtype=torch.int32, device="cuda")
    hip.hipMemcpy(ctypes.c_void_p(h.data_ptr()), ctypes.c_void_p(t.get_option("debug_ctl_ptr")), ctypes.c_size_t(160), 3)
    torch.cuda.synchronize()
    w = h.cpu().numpy().view(np.uint32)
    assert int(w[3]) == 12 and int(w[38]) == 64 - 12 and int(w[36]) == 0, w[:40]   # launch clock stopped at 4 + 8, 12 launches of the pause gone
    last = sc.pinhole_rays(1280, 800, eye0 + np.array([0.02 * 20, 0.01 * 20, 0.0]), centre, 45.0)
    assert_hits_equal(out.cpu().numpy().view(rc.HIT_DT), o.trace(last, nthreads=16), "the last of 20 frames enqueued without waiting")
    t.free()
