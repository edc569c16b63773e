"""Multi-GPU drivers: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

Only the parts of the path that shard get a collective:

* closest_hit / any_hit batches: rays are independent, the scene is a few MB -> replicas; `shard_range`
  splits the ray array, no data-path collective.
* view_factors (src/kernels.jl:74-104): every (source primitive, ray) pair is independent and draws its
  randomness from Philox keyed by (seed; ray, source), so the job can be cut either way and the result is
  bit-identical to the single-GPU matrix.  Sources are addressed in METADATA order (RC_VF_SOURCES_BY_METADATA):
  with the view-factor convention metadata = a permutation of 1..N, a contiguous source range is a contiguous block of
  FINAL matrix rows, so no partition needs a row permutation or a second N x N buffer.
    - mode="rows_sharded": rank g owns rows [gN/G, (g+1)N/G), shoots all their rays into its own block and the result STAYS
      sharded (block + row index): no collective at all -- the layout a row-parallel consumer (e.g. a radiosity solve)
      wants, and the only one whose cost does not include 4 N^2 bytes over xGMI.
    - mode="rows": the same compute, then the blocks are collected on rank `dst`: the root traces straight into its
      slice of the final matrix and receives every peer's block straight into that peer's slice (point-to-point, all peers
      at once: xGMI is point-to-point, the G-1 transfers use the root's G-1 links concurrently) -- each byte crosses xGMI
      once, nothing is padded, copied or permuted afterwards.
    - mode="rays" (the north-star wording): every rank shoots rays [r0, r1) of ALL sources into a full N x N accumulator
      and the accumulators are summed on `dst` with RCCL reduce.  The job is cut into row chunks: while chunk k is being
      traced the reduce of chunk k-1 is in flight (async_op), so all but the last chunk's exchange hides behind tracing
      as far as the link budget allows (DESIGN.md section 5 has the byte / time budget).
  Metadata that are not a permutation of 1..N (duplicates, gaps) take the general path: rows addressed by primitive,
  one index_add by metadata on the root.
* get_illumination: the ray grid is cut into contiguous ranges, each rank histograms its range, one
  reduce(SUM) of N floats (exact: integer-valued f32 counts).

`compute` hooks let the CPU tests (gloo, world_size 2) drive the same sharding + collective logic with a
stand-in for the GPU kernels.
"""
import numpy as np

from . import _capi
from ._capi import check, lib, ptr

VF_ROW_BY_PRIMITIVE, VF_SOURCES_BY_METADATA = 1, 2


def shard_range(n, rank, world):
    """Contiguous, balanced split of range(n): returns (begin, end) of `rank`'s shard."""
    return (n * rank) // world, (n * (rank + 1)) // world


def _dist():
    import torch.distributed as dist
    return dist


def _ranks(group):
    """(world size, this process's rank IN THE GROUP, its global rank).  Sharding uses the group rank; `dst` arguments are
    global ranks, as everywhere in torch.distributed."""
    dist = _dist()
    if not dist.is_initialized():
        return 1, 0, 0
    return dist.get_world_size(group), dist.get_rank(group), dist.get_rank()


def _global_rank(group, group_rank):
    dist = _dist()
    if group is None or not dist.is_initialized():
        return group_rank
    return dist.get_global_rank(group, group_rank)


def _gpu_view_factors(tlas, rays_per_triangle, seed):
    def compute(local, src, rays, row_stride, col_stride, row_offset, addressing):
        import torch
        flags = {None: 0, "primitive": VF_ROW_BY_PRIMITIVE, "metadata": VF_SOURCES_BY_METADATA}[addressing]
        check(lib().rc_view_factors_device(tlas._h, int(rays_per_triangle), int(seed), src[0], src[1], rays[0], rays[1],
                                           ptr(local.data_ptr()), row_stride, col_stride, row_offset, flags,
                                           ptr(torch.cuda.current_stream().cuda_stream) or None))
    return compute


def view_factors_distributed(tlas, rays_per_triangle=10000, seed=0, mode="rows", group=None, dst=0, n_prims=None,
                             compute=None, device=None, prim_meta=None, chunks=None):
    """view_factors sharded over the ranks of `group`.  Returns, on global rank `dst`, an int32 torch tensor whose
    uint32 view is the N x N matrix M[src_meta-1, hit_meta-1] (row-major; Julia's Matrix is its transpose in
    memory); other ranks return None.  mode="rows_sharded" returns (block, row_index) on EVERY rank instead: block[r]
    is matrix row row_index[r].  `compute(local, (s0,s1), (r0,r1), row_stride, col_stride, row_offset, addressing)` must
    ACCUMULATE into `local`; addressing = "metadata" (source positions and rows in metadata order), "primitive" (flat
    primitive order) or None (sources by primitive, rows by metadata); the default launches the HIP kernel through the C ABI.
    prim_meta: metadata of the flat (Morton-sorted) primitive array (default: read back from the scene).
    chunks: row chunks of the pipelined "rays" exchange (default: about 256 MiB of matrix per chunk, at least 2 x world)."""
    import torch
    dist = _dist()
    world, rank, me = _ranks(group)
    n = int(n_prims if n_prims is not None else tlas.n_primitives())
    rpt = int(rays_per_triangle)
    if device is None:
        device = torch.device("cuda", tlas.device)
    if compute is None:
        compute = _gpu_view_factors(tlas, rpt, seed)
    if prim_meta is None:
        prim_meta = tlas._prims()["meta"]
    meta = np.asarray(prim_meta).astype(np.int64)
    order = np.argsort(meta, kind="stable")                 # position in metadata order -> flat primitive index
    is_perm = len(meta) == n and np.array_equal(meta[order], np.arange(1, n + 1))
    if mode not in ("rays", "rows", "rows_sharded"):
        raise ValueError("mode must be 'rays', 'rows' or 'rows_sharded'")

    if mode == "rows_sharded":
        s0, s1 = shard_range(n, rank, world)
        local = torch.zeros((s1 - s0) * n, dtype=torch.int32, device=device)
        compute(local, (s0, s1), (0, rpt), n, 1, s0, "metadata")
        # row r of the block is the source at position s0 + r of the metadata order: matrix row meta - 1 (= s0 + r for a permutation)
        return local.view(s1 - s0, n), meta[order[s0:s1]] - 1

    if not is_perm:
        return _view_factors_general(dist, group, world, rank, me, dst, mode, n, rpt, compute, device, meta)

    if mode == "rows":
        out = torch.zeros(n * n if me == dst else 0, dtype=torch.int32, device=device)
        s0, s1 = shard_range(n, rank, world)
        if me == dst:
            compute(out[s0 * n:s1 * n], (s0, s1), (0, rpt), n, 1, s0, "metadata")    # the root traces into its slice of the result
            if world > 1:
                ops = []
                for r in range(world):
                    if r == rank:
                        continue
                    a, b = shard_range(n, r, world)
                    ops.append(dist.P2POp(dist.irecv, out[a * n:b * n], _global_rank(group, r), group))
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            return out.view(n, n)
        local = torch.zeros((s1 - s0) * n, dtype=torch.int32, device=device)
        compute(local, (s0, s1), (0, rpt), n, 1, s0, "metadata")
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, local, dst, group)]):
            w.wait()
        return None

    # mode == "rays": a full accumulator per rank, reduced chunk by chunk while the next chunk is traced
    r0, r1 = shard_range(rpt, rank, world)
    local = torch.zeros(n * n, dtype=torch.int32, device=device)
    if chunks is None:
        chunks = max(2 * world, (4 * n * n + (256 << 20) - 1) // (256 << 20)) if world > 1 else 1
    chunks = max(1, min(int(chunks), n))
    pending = []
    for k in range(chunks):
        c0, c1 = shard_range(n, k, chunks)
        slab = local[c0 * n:c1 * n]
        compute(slab, (c0, c1), (r0, r1), n, 1, c0, "metadata")
        if world > 1:
            pending.append(dist.reduce(slab, dst=dst, op=dist.ReduceOp.SUM, group=group, async_op=True))
    for w in pending:
        w.wait()
    return local.view(n, n) if me == dst else None


def _gpu_view_factor_totals(tlas, rays_per_triangle, seed):
    def compute(local, src, rays):
        import torch
        n = local.numel() // 2
        base = local.data_ptr()
        check(lib().rc_view_factor_totals_device(tlas._h, int(rays_per_triangle), int(seed), src[0], src[1], rays[0], rays[1], ptr(base), ptr(base + 8 * n),
                                                 ptr(torch.cuda.current_stream().cuda_stream) or None))
    return compute


def view_factor_totals_distributed(tlas, rays_per_triangle=10000, seed=0, group=None, dst=0, n_prims=None, compute=None, device=None):
    """The per-triangle totals of view_factors -- received[j] = column sum j, emitted[i] = row sum i of the N x N matrix, the quantities the
    reference's users read off it (docs/src/viewfactors_content.md:62-68) -- with the RAYS sharded over the ranks of `group`: rank r shoots
    ray indices shard_range(rays_per_triangle, r, world) of every source into ONE int64 tensor of 2 N elements (received | emitted) and a
    single dist.reduce (ncclReduce over xGMI under backend "nccl" = RCCL; 0.8 MB at C5) sums them on global rank `dst`.  No N x N array on
    any rank.  Returns (received, emitted) as uint64 numpy vectors on `dst`, None elsewhere.  `compute(local, (s0, s1), (r0, r1))` must
    ACCUMULATE the totals of those sources / rays into `local`; the default launches the HIP kernel through the C ABI."""
    import torch
    dist = _dist()
    world, rank, me = _ranks(group)
    n = int(n_prims if n_prims is not None else tlas.n_primitives())
    rpt = int(rays_per_triangle)
    if device is None:
        device = torch.device("cuda", tlas.device)
    if compute is None:
        compute = _gpu_view_factor_totals(tlas, rpt, seed)
    local = torch.zeros(2 * n, dtype=torch.int64, device=device)
    r0, r1 = shard_range(rpt, rank, world)
    if r1 > r0:
        compute(local, (0, n), (r0, r1))
    if world > 1:
        dist.reduce(local, dst=dst, op=dist.ReduceOp.SUM, group=group)
    if me != dst:
        return None
    host = local.cpu().numpy().view(np.uint64)
    return host[:n].copy(), host[n:].copy()


def _populate(base, byte_range, n_threads):
    """MADV_POPULATE_WRITE (Linux 5.14+) over [base + a, base + b), page aligned, split over n_threads; best effort."""
    import ctypes
    import threading
    a, b = byte_range
    a -= (base + a) % 4096
    if b <= a:
        return
    try:
        libc = ctypes.CDLL(None, use_errno=True)
        libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    except (OSError, AttributeError):
        return
    step = -(-(b - a) // n_threads)
    step += -step % 4096
    ths = [threading.Thread(target=libc.madvise, args=(base + s, min(step, b - s), 23)) for s in range(a, b, step)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()


class SharedHostMatrix:
    """An N x N uint32 matrix, column-major like Julia's Matrix, in shared memory (/dev/shm) that every rank of the group maps.  Creating
    it is collective and costs what 4 N^2 bytes of fresh shared-memory pages cost (C5: 10 GB, ~3 s from one rank, split G ways with G
    ranks); a solver that calls view_factors repeatedly creates it ONCE and passes it as `out=` -- the calls then cost the PCIe time of
    each rank's row block.  `array` is the np.memmap (every rank has the whole matrix mapped; rank `dst` is the one that returns it)."""

    def __init__(self, n, group=None, dst=0):
        import os
        dist = _dist()
        world, rank, me = _ranks(group)
        self.n, self.group, self.dst = int(n), group, dst
        name = [None]
        if me == dst:
            name[0] = f"/dev/shm/raycore_vf_{os.getpid()}_{np.random.default_rng().integers(1 << 62):x}"
            with open(name[0], "wb") as f:
                f.truncate(max(1, 4 * self.n * self.n))
        if world > 1:
            dist.broadcast_object_list(name, src=dst, group=group)
        try:
            self.array = np.memmap(name[0], dtype=np.uint32, mode="r+", shape=(self.n, self.n), order="F")
            # fault the pages in before the copies arrive: every rank its own slice of the file, a few threads per rank (shared-memory
            # pages cost ~0.15 s per GB to allocate from one thread; ctypes releases the GIL during the madvise calls)
            _populate(self.array.ctypes.data, shard_range(4 * self.n * self.n, rank, world), max(1, min(8, (os.cpu_count() or 8) // max(world, 1))))
            if world > 1:
                dist.barrier(group=group)
        finally:
            if me == dst:
                os.unlink(name[0])  # the mappings keep the pages alive


def view_factors_host_matrix(tlas, rays_per_triangle=10000, seed=0, group=None, dst=0, n_prims=None, compute_rows=None, out=None):
    """view_factors as the API returns it -- a HOST N x N uint32 matrix, column-major (src/kernels.jl:74-78) -- from one process per GPU:
    rank `dst` creates the matrix in shared memory (/dev/shm), every rank maps it and brings ITS block of rows home over its own PCIe
    link (rc_view_factors_rows_host: row chunks traced while the finished ones are copied), so G links run in parallel, nothing
    crosses xGMI and the only collective is the barrier at the end.  This is the partition that serves the API's return value: at
    C5 the matrix is 10 GB -- 0.18 s over one link, whatever the tracing costs -- against 23 ms per rank with eight.
    Returns the matrix (a np.memmap, order F; the file is already unlinked) on `dst`, None elsewhere.
    out: a SharedHostMatrix created earlier by every rank of the group -- reuses its pages (every element is overwritten).
    compute_rows(out, (r0, r1)): stand-in for the device call in the CPU tests (fills rows [r0, r1) of `out`)."""
    dist = _dist()
    world, rank, me = _ranks(group)
    n = int(n_prims if n_prims is not None else tlas.n_primitives())
    if out is None:
        out = SharedHostMatrix(n, group, dst)
    elif out.n != n:
        raise ValueError(f"out is a {out.n} x {out.n} matrix, the scene has {n} primitives")
    m = out.array
    r0, r1 = shard_range(n, rank, world)
    if compute_rows is not None:
        compute_rows(m, (r0, r1))
    else:
        check(lib().rc_view_factors_rows_host(tlas._h, int(rays_per_triangle), int(seed), r0, r1, m.ctypes.data_as(_capi.C.c_void_p), n))
    if world > 1:
        dist.barrier(group=group)
    return m if me == dst else None


def _view_factors_general(dist, group, world, rank, me, dst, mode, n, rpt, compute, device, meta):
    """Metadata with duplicates or gaps: rows cannot be addressed by metadata position, so rows travel in primitive order and the
    root folds them by metadata with one index_add (duplicates accumulate, as result[src_meta, :] does in the reference)."""
    import torch
    rows = torch.as_tensor(meta - 1, device=device)
    if mode == "rays":
        r0, r1 = shard_range(rpt, rank, world)
        local = torch.zeros(n * n, dtype=torch.int32, device=device)
        compute(local, (0, n), (r0, r1), n, 1, 0, None)  # row-major [src_meta-1][hit_meta-1]
        if world > 1:
            dist.reduce(local, dst=dst, op=dist.ReduceOp.SUM, group=group)
        return local.view(n, n) if me == dst else None
    s0, s1 = shard_range(n, rank, world)
    local = torch.zeros((s1 - s0) * n, dtype=torch.int32, device=device)
    compute(local, (s0, s1), (0, rpt), n, 1, s0, "primitive")
    if me != dst:
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, local, dst, group)]):
            w.wait()
        return None
    out = torch.zeros(n, n, dtype=torch.int32, device=device)
    valid = (rows >= 0) & (rows < n)
    for r in range(world):
        a, b = shard_range(n, r, world)
        if r == rank:
            part = local
        else:
            part = torch.empty((b - a) * n, dtype=torch.int32, device=device)
            for w in dist.batch_isend_irecv([dist.P2POp(dist.irecv, part, _global_rank(group, r), group)]):
                w.wait()
        sel = valid[a:b]
        out.index_add_(0, rows[a:b][sel], part.view(b - a, n)[sel])
        del part
    return out


def get_illumination_distributed(tlas, viewdir, grid_size=1000, group=None, dst=0, n_prims=None, compute=None, device=None):
    """get_illumination with the ray grid sharded over ranks and one reduce(SUM) of the N-float histogram."""
    import torch
    dist = _dist()
    world, rank, me = _ranks(group)
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
    return local[:n] if me == dst else None


def trace_sharded(tlas, rays, mode="closest", group=None):
    """Replica tracing: this rank traces its contiguous shard of `rays` (RAY_DT array) and returns
    ((begin, end), hits) -- no collective; the caller concatenates shards if it needs the whole batch."""
    world, rank, _ = _ranks(group)
    b, e = shard_range(len(rays), rank, world)
    return (b, e), tlas.trace(rays[b:e], mode=mode)
