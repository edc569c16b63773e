"""world_size-2 gloo tests (CPU) of the multi-GPU sharding + collective logic (raycore.jl_amd/distributed.py).
The GPU kernel is replaced by the CPU oracle through the `compute` hook; what is under test is the
partitioning, the padding/gather/reduce and that the sharded result equals the single-process matrix."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _scene(po, rc, dup_meta=False):
    sc = rc.scenes
    verts = np.concatenate([sc.fan_sphere(8, 5, radius=0.5), sc.box_room((-1.5, -1.5, -1.5), (1.5, 1.5, 1.5), 1)])
    n = len(verts)
    meta = np.arange(1, n + 1, dtype=np.uint32)
    if dup_meta:  # metadata that are not a permutation of 1..N: groups of faces share a row, one id is out of range
        meta = (meta + 1) // 2
        meta[3] = n + 7
    s = po.Scene()
    b = s.add_blas(verts, meta)
    s.add_instance(b)
    return s.build(), n


def _worker(rank, world, port, mode, q, variant="plain"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    import raycore_jl_amd as rc
    from raycore_jl_amd import distributed as rd
    from oracle import pyoracle as po
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o, n = _scene(po, rc, dup_meta=(variant == "dup"))
        rpt, seed = 64, 99

        meta = o.blas_prims["meta"].astype(np.int64)
        order = np.argsort(meta, kind="stable")
        group, dst = None, 0
        if variant == "subgroup":  # ranks 1 and 2 of a 3-process world: group rank != global rank, dst is the GLOBAL rank 2
            group, dst = dist.new_group([1, 2]), 2

        def compute(local, src, rays, row_stride, col_stride, row_offset, addressing):
            """Stand-in for rc_view_factors_device: the oracle's rays for one source at a time (rco_view_factor_row), accumulated with
            the kernel's addressing rules (RC_VF_SOURCES_BY_METADATA / RC_VF_ROW_BY_PRIMITIVE / metadata rows)."""
            assert (row_stride, col_stride) == (n, 1)
            rows = local.numpy().view(np.uint32).reshape(-1, n)
            for pos in range(src[0], src[1]):
                prim = order[pos] if addressing == "metadata" else pos
                row = pos if addressing in ("metadata", "primitive") else meta[prim] - 1
                if not (1 <= meta[prim] <= n):
                    continue  # view_factors! would index out of bounds; the kernel drops such sources
                rows[row - row_offset] += o.view_factor_row(rpt, int(prim), seed=seed, rays=rays)

        if variant == "subgroup" and rank == 0:
            res = "not in the group"
        elif mode in ("rays", "rows"):
            out = rd.view_factors_distributed(None, rpt, seed, mode=mode, n_prims=n, compute=compute, device=torch.device("cpu"), prim_meta=meta,
                                              group=group, dst=dst, chunks=5 if mode == "rays" else None)
            res = None if out is None else out.numpy().view(np.uint32).copy()
        elif mode == "host_matrix":
            def compute_rows(out, rng):
                """Stand-in for rc_view_factors_rows_host: matrix rows [r0, r1) of the shared host matrix (column-major), from the oracle."""
                out[rng[0]:rng[1], :] = 0  # like the real call, every element of the rank's rows is written
                for prim in range(n):
                    r = int(meta[prim]) - 1
                    if rng[0] <= r < rng[1]:
                        out[r, :] += o.view_factor_row(rpt, int(prim), seed=seed)
            out = rd.view_factors_host_matrix(None, rpt, seed, n_prims=n, compute_rows=compute_rows)
            first = None if out is None else np.array(out)  # a copy: the mapping dies with the worker
            assert out is None or (out.flags["F_CONTIGUOUS"] and not [f for f in os.listdir("/dev/shm") if f.startswith(f"raycore_vf_{os.getpid()}_")])
            # a matrix created once and filled repeatedly (a solver's loop): stale contents are overwritten, nothing is re-created
            shared = rd.SharedHostMatrix(n)
            shared.array[:] = 0xDEADBEEF
            dist.barrier()
            for _ in range(2):
                again = rd.view_factors_host_matrix(None, rpt, seed, n_prims=n, compute_rows=compute_rows, out=shared)
            assert (again is None) == (out is None) and (again is None or again is shared.array)
            res = None if out is None else (first, np.array(again))
        elif mode == "totals":
            def compute_totals(local, src, rays):
                """Stand-in for rc_view_factor_totals_device: received[hit_meta - 1] and emitted[src_meta - 1] of the oracle's rays."""
                acc = local.numpy().view(np.uint64)
                for prim in range(src[0], src[1]):
                    if not (1 <= meta[prim] <= n):
                        continue
                    row = o.view_factor_row(rpt, int(prim), seed=seed, rays=rays).astype(np.uint64)
                    acc[:n] += row
                    acc[n + meta[prim] - 1] += row.sum()
            out = rd.view_factor_totals_distributed(None, rpt, seed, n_prims=n, compute=compute_totals, device=torch.device("cpu"), group=group, dst=dst)
            res = None if out is None else (out[0].copy(), out[1].copy())
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


def _run(world, mode, variant, port_offset):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + port_offset
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, q, variant)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


@pytest.mark.parametrize("mode", ["rays", "rows"])
def test_metadata_with_duplicates_takes_the_general_path(oracle, mode):
    """Metadata that are not a permutation of 1..N (shared rows, an out-of-range id): rows travel in primitive order and the root
    folds them by metadata; the result is the reference's matrix (result[src_meta, hit_meta], out-of-range ids dropped)."""
    import raycore_jl_amd as rc
    results = _run(2, mode, "dup", 10 + (mode == "rows"))
    o, n = _scene(oracle, rc, dup_meta=True)
    want = o.view_factors(64, seed=99)
    assert results[1] is None and np.array_equal(results[0], want) and want.sum() > 0


@pytest.mark.parametrize("mode", ["rays", "rows"])
def test_subgroup_uses_group_ranks_for_shards_and_a_global_rank_for_dst(oracle, mode):
    """dist.get_rank(group) is a group-local rank; `dst` (and every peer of a send / recv) is a global rank.  Ranks 1 and 2
    of a three-process world form the group, the matrix lands on global rank 2."""
    import raycore_jl_amd as rc
    results = _run(3, mode, "subgroup", 20 + (mode == "rows"))
    o, n = _scene(oracle, rc)
    assert results[0] == "not in the group" and results[1] is None
    assert np.array_equal(results[2], o.view_factors(64, seed=99))


@pytest.mark.parametrize("variant,world", [("plain", 2), ("dup", 2), ("subgroup", 3)])
def test_view_factor_totals_rays_sharded_one_reduce(oracle, variant, world):
    """view_factor_totals_distributed: every rank shoots its share of the ray indices of EVERY source into a 2 N vector, one reduce to
    dst; the result is the column / row sums of the single-process matrix -- also for metadata with duplicates and an out-of-range id,
    and for a sub-group whose dst is a global rank."""
    import raycore_jl_amd as rc
    results = _run(world, "totals", variant, 40 + {"plain": 0, "dup": 1, "subgroup": 2}[variant])
    o, n = _scene(oracle, rc, dup_meta=(variant == "dup"))
    want = o.view_factors(64, seed=99)
    dst = 2 if variant == "subgroup" else 0
    for r, res in results.items():
        if r != dst:
            assert res is None or res == "not in the group"
    recv, emit = results[dst]
    assert np.array_equal(recv, want.sum(axis=0, dtype=np.uint64)) and np.array_equal(emit, want.sum(axis=1, dtype=np.uint64)) and recv.sum() > 0


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a rank environment starts its own two rank processes (before anything touches a GPU) and
    relays rank 0's ONE JSON line; under torch.distributed.run it takes the ranks it is given.  --dry-run: no GPU work."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1", "--res", "64"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["scaling"] == "weak" and out["value"] > 0
    port = 29000 + os.getpid() % 400
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1", "--res", "64"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2
    # a rank count that contradicts the environment is an error, not a silent single-GPU run
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True, timeout=120,
                       env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0


def test_host_matrix_in_shared_memory_world2(oracle):
    """view_factors_host_matrix: rank 0 creates the N x N host matrix in /dev/shm, both ranks map it and fill their own rows (here the
    oracle stands in for rc_view_factors_rows_host), barrier, the file is unlinked; rank 0 holds the reference's column-major matrix."""
    import raycore_jl_amd as rc
    results = _run(2, "host_matrix", "plain", 30)
    o, n = _scene(oracle, rc)
    assert results[1] is None
    want = o.view_factors(64, seed=99)
    fresh, reused = results[0]
    assert np.array_equal(fresh, want) and fresh.sum() > 0
    assert np.array_equal(reused, want)   # through a SharedHostMatrix created once, prefilled with garbage


def test_shard_range_covers_everything():
    from raycore_jl_amd.distributed import shard_range
    for n in (0, 1, 7, 64, 50028):
        for world in (1, 2, 3, 8):
            parts = [shard_range(n, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            assert max(e - b for b, e in parts) - min(e - b for b, e in parts) <= 1
