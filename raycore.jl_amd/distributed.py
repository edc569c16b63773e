"""Multi-GPU drivers: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

Only the parts of the path that shard get a collective:

* closest_hit / any_hit batches: rays are independent, the scene is a few MB -> replicas; `shard_range`
  splits the ray array, no data-path collective.
* view_factors (src/kernels.jl:74-104): every (source primitive, ray) pair is independent and draws its
  randomness from Philox keyed by (seed; ray, source), so the job can be cut either way and the result is
  bit-identical to the single-GPU matrix:
    - mode="rays"  (the north-star wording): every rank shoots rays [r0, r1) of ALL source primitives into a
      full N x N accumulator, then ONE reduce(SUM) of the accumulators over RCCL.  Message = 4 N^2 bytes.
    - mode="rows_sharded": like "rows" below but the result STAYS sharded (each rank returns its own row block and the
      metadata of its rows): no collective at all -- the layout a row-parallel consumer (e.g. a radiosity solve) wants,
      and the only one whose cost does not grow with N^2 bytes over xGMI.
    - mode="rows": rank g owns source primitives [s0, s1) and shoots all their rays into its own
      (s1 - s0) x N row block; rows are disjoint (result[src, :] is written only by src, :85-97), so the
      exchange is a gather of row blocks -- each byte crosses xGMI once, the 7 peers use the root's 7
      distinct links concurrently.  Preferred when N is large (N = 50k: 10 GB matrix).
* get_illumination: the ray grid is cut into contiguous ranges, each rank histograms its range, one
  reduce(SUM) of N floats (exact: integer-valued f32 counts).

`compute` hooks let the CPU tests (gloo, world_size 2) drive the same sharding + collective logic with a
stand-in for the GPU kernels.
"""
import numpy as np

from . import _capi
from ._capi import check, lib, ptr


def shard_range(n, rank, world):
    """Contiguous, balanced split of range(n): returns (begin, end) of `rank`'s shard."""
    return (n * rank) // world, (n * (rank + 1)) // world


def _dist():
    import torch.distributed as dist
    return dist


def _gpu_view_factors(tlas, rays_per_triangle, seed):
    def compute(local, src, rays, row_stride, col_stride, row_offset, by_prim):
        import torch
        check(lib().rc_view_factors_device(tlas._h, int(rays_per_triangle), int(seed), src[0], src[1], rays[0], rays[1],
                                           ptr(local.data_ptr()), row_stride, col_stride, row_offset, 1 if by_prim else 0,
                                           ptr(torch.cuda.current_stream().cuda_stream) or None))
    return compute


def view_factors_distributed(tlas, rays_per_triangle=10000, seed=0, mode="rows", group=None, dst=0, n_prims=None,
                             compute=None, device=None, prim_meta=None):
    """view_factors sharded over the ranks of `group`.  Returns, on rank `dst`, an int32 torch tensor whose
    uint32 view is the N x N matrix M[src_meta-1, hit_meta-1] (row-major; Julia's Matrix is its transpose in
    memory); other ranks return None.  mode="rows_sharded" returns (block, row_index) on EVERY rank instead: block[r]
    is matrix row row_index[r].  `compute(local, (s0,s1), (r0,r1), row_stride, col_stride, row_offset,
    by_prim)` must ACCUMULATE into `local`; the default launches the HIP kernel through the C ABI.
    prim_meta: metadata of the flat (Morton-sorted) primitive array (default: read back from the scene)."""
    import torch
    dist = _dist()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = int(n_prims if n_prims is not None else tlas.n_primitives())
    if device is None:
        device = torch.device("cuda", tlas.device)
    if compute is None:
        compute = _gpu_view_factors(tlas, rays_per_triangle, seed)
    if mode == "rays":
        r0, r1 = shard_range(int(rays_per_triangle), rank, world)
        local = torch.zeros(n * n, dtype=torch.int32, device=device)
        compute(local, (0, n), (r0, r1), n, 1, 0, False)  # row-major [src_meta-1][hit_meta-1]
        if world > 1:
            dist.reduce(local, dst=dst, op=dist.ReduceOp.SUM, group=group)
        return local.view(n, n) if rank == dst else None
    if mode == "rows_sharded":
        s0, s1 = shard_range(n, rank, world)
        local = torch.zeros((s1 - s0) * n, dtype=torch.int32, device=device)
        compute(local, (s0, s1), (0, int(rays_per_triangle)), n, 1, s0, True)
        if prim_meta is None:
            prim_meta = tlas._prims()["meta"]
        # row r of the block belongs to matrix row prim_meta[s0 + r] - 1
        return local.view(s1 - s0, n), np.asarray(prim_meta)[s0:s1].astype(np.int64) - 1
    if mode == "rows":
        s0, s1 = shard_range(n, rank, world)
        rows_max = max(shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world))
        local = torch.zeros(rows_max * n, dtype=torch.int32, device=device)  # padded to the largest shard
        # rows are indexed by the source's position in the Morton-sorted primitive array, so a contiguous
        # source range owns a contiguous block; the root scatters rows to their metadata slot once.
        compute(local, (s0, s1), (0, int(rays_per_triangle)), n, 1, s0, True)
        if rank == dst:
            if world > 1:
                parts = [torch.empty_like(local) for _ in range(world)]
                dist.gather(local, gather_list=parts, dst=dst, group=group)
            else:
                parts = [local]
            if prim_meta is None:
                prim_meta = tlas._prims()["meta"]
            rows = torch.as_tensor(np.asarray(prim_meta).astype(np.int64) - 1, device=device)
            out = torch.zeros(n, n, dtype=torch.int32, device=device)
            for r, p in enumerate(parts):
                a, b = shard_range(n, r, world)
                out.index_add_(0, rows[a:b], p[:(b - a) * n].view(b - a, n))  # index_add: duplicate metadata accumulates, like the reference
            return out
        dist.gather(local, gather_list=None, dst=dst, group=group)
        return None
    raise ValueError("mode must be 'rays', 'rows' or 'rows_sharded'")


def get_illumination_distributed(tlas, viewdir, grid_size=1000, group=None, dst=0, n_prims=None, compute=None, device=None):
    """get_illumination with the ray grid sharded over ranks and one reduce(SUM) of the N-float histogram."""
    import torch
    dist = _dist()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = int(n_prims if n_prims is not None else tlas.n_primitives())
    if device is None:
        device = torch.device("cuda", tlas.device)
    b, e = shard_range(int(grid_size) * int(grid_size), rank, world)
    local = torch.zeros(max(n, 1), dtype=torch.float32, device=device)
    if compute is None:
        vd = np.ascontiguousarray(viewdir, dtype=np.float32)
        check(lib().rc_get_illumination_device(tlas._h, ptr(vd), int(grid_size), b, e, ptr(local.data_ptr()),
                                               ptr(torch.cuda.current_stream().cuda_stream) or None))
    else:
        compute(local, (b, e))
    if world > 1:
        dist.reduce(local, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return local[:n] if rank == dst else None


def trace_sharded(tlas, rays, mode="closest", group=None):
    """Replica tracing: this rank traces its contiguous shard of `rays` (RAY_DT array) and returns
    ((begin, end), hits) -- no collective; the caller concatenates shards if it needs the whole batch."""
    dist = _dist()
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    b, e = shard_range(len(rays), rank, world)
    return (b, e), tlas.trace(rays[b:e], mode=mode)
