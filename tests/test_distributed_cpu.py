"""world_size-2 gloo tests (CPU) of the multi-GPU sharding + collective logic (raycore.jl_amd/distributed.py).
The GPU kernel is replaced by the CPU oracle through the `compute` hook; what is under test is the
partitioning, the padding/gather/reduce and that the sharded result equals the single-process matrix."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scene(po, rc):
    sc = rc.scenes
    verts = np.concatenate([sc.fan_sphere(8, 5, radius=0.5), sc.box_room((-1.5, -1.5, -1.5), (1.5, 1.5, 1.5), 1)])
    n = len(verts)
    s = po.Scene()
    b = s.add_blas(verts, np.arange(1, n + 1, dtype=np.uint32))
    s.add_instance(b)
    return s.build(), n


def _worker(rank, world, port, mode, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    import raycore_jl_amd as rc
    from raycore_jl_amd import distributed as rd
    from oracle import pyoracle as po
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o, n = _scene(po, rc)
        rpt, seed = 64, 99

        meta = o.blas_prims["meta"]

        def compute(local, src, rays, row_stride, col_stride, row_offset, by_prim):
            assert (row_stride, col_stride) == (n, 1)
            m = o.view_factors(rpt, seed=seed, src=src, rays=rays)  # m[src_meta-1, hit_meta-1]
            rows = local.numpy().view(np.uint32).reshape(-1, n)
            for p_idx in range(src[0], src[1]):
                row = p_idx if by_prim else meta[p_idx] - 1
                rows[row - row_offset] += m[meta[p_idx] - 1]

        if mode in ("rays", "rows"):
            out = rd.view_factors_distributed(None, rpt, seed, mode=mode, n_prims=n, compute=compute, device=torch.device("cpu"), prim_meta=meta)
            res = None if out is None else out.numpy().view(np.uint32).copy()
        elif mode == "rows_sharded":
            block, rows = rd.view_factors_distributed(None, rpt, seed, mode=mode, n_prims=n, compute=compute, device=torch.device("cpu"), prim_meta=meta)
            res = (block.numpy().view(np.uint32).copy(), rows)
        else:
            grid = 40
            rays = o.ray_grid([0.2, 0.3, 1.0], grid)

            def compute_illum(local, rng):
                hits = o.trace(rays[rng[0]:rng[1]])
                metas = o.blas_prims["meta"][hits["primitive_id"][hits["hit"] == 1]]
                local.numpy()[:n] += np.bincount(metas - 1, minlength=n).astype(np.float32)

            out = rd.get_illumination_distributed(None, [0.2, 0.3, 1.0], grid, n_prims=n, compute=compute_illum, device=torch.device("cpu"))
            res = None if out is None else out.numpy().copy()
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["rays", "rows", "rows_sharded", "illum"])
def test_sharded_drivers_world2(oracle, mode):
    import torch.multiprocessing as mp
    import raycore_jl_amd as rc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + {"rays": 0, "rows": 1, "illum": 2, "rows_sharded": 3}[mode]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    o, n = _scene(oracle, rc)
    if mode == "rows_sharded":  # every rank keeps its own rows; together they are the whole matrix
        want = o.view_factors(64, seed=99)
        full = np.zeros_like(want)
        for r in (0, 1):
            block, rows = results[r]
            full[rows] += block
        assert np.array_equal(full, want) and len(results[0][1]) + len(results[1][1]) == n
        return
    assert results[1] is None
    if mode == "illum":
        assert np.array_equal(results[0], o.get_illumination([0.2, 0.3, 1.0], 40))
    else:
        want = o.view_factors(64, seed=99)
        assert results[0].shape == (n, n) and np.array_equal(results[0], want) and want.sum() > 0


def test_shard_range_covers_everything():
    from raycore_jl_amd.distributed import shard_range
    for n in (0, 1, 7, 64, 50028):
        for world in (1, 2, 3, 8):
            parts = [shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            assert max(e - b for b, e in parts) - min(e - b for b, e in parts) <= 1
