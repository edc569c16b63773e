# RaycoreMI355X.jl -- the ccall binding a Raycore.jl maintainer would add to reach libraycore_mi355x.so.
#
# NOT EXECUTABLE IN THE BUILD IMAGE (no julia binary there); written against include/raycore_mi355x.h and the
# reference API (file:line citations are relative to the Raycore.jl repo).  It implements the AbstractAccel
# contract (src/Raycore.jl:14-49) the same way Lava.HWTLAS does: a mutable accel with push!/delete!/update_*!/
# sync!, an adapted form handed to consumers per dispatch, and batched trace entry points
# (docs/src/hw_acceleration.md:141-146 is the precedent for batched dispatch behind this contract).
module RaycoreMI355X

import Raycore
import Raycore: AbstractAccel, AbstractAdaptedAccel, TLASHandle, Triangle, Bounds3, RTRay, RTHitResult,
                InstanceDescriptor, Mat3x4f, mat4_to_mat3x4, empty_triangle
import Adapt
using GeometryBasics, StaticArrays

const LIB = get(ENV, "RAYCORE_MI355X_LIB", "libraycore_mi355x.so")

struct MI355XBackend
    device::Cint
end
MI355XBackend() = MI355XBackend(0)

last_error() = unsafe_string(ccall((:rc_last_error, LIB), Cstring, ()))
# roctx ranges around a caller's own phases (every entry point of the library is a range of its own): `profile_range("shade") do ... end` (not `range`: that would shadow Base.range inside the module)
range_push(name::AbstractString) = ccall((:rc_range_push, LIB), Cint, (Cstring,), name)
range_pop() = ccall((:rc_range_pop, LIB), Cint, ())
ranges_enabled() = ccall((:rc_ranges_enabled, LIB), Cint, ()) != 0
function profile_range(f, name::AbstractString)
    range_push(name)
    try
        return f()
    finally
        range_pop()
    end
end
# every non-zero status becomes ErrorException, the type the reference's tests expect
# (test/test_tlas_stress.jl:585-617: @test_throws ErrorException update_transform!(tlas, deleted_handle, ...))
check(status::Cint) = status == 0 ? nothing : error(last_error())

"Mutable accel: plays the role of Raycore.TLAS{Backend} (src/instanced-bvh.jl:261-310)."
mutable struct MI355XTLAS <: AbstractAccel
    backend::MI355XBackend
    ptr::Ptr{Cvoid}
    prims::Vector{Triangle{UInt32}}      # host copy of all_blas_prims (Morton-sorted), refreshed after a rebuild
    prims_valid::Bool
    # Triangle{TMeta} for any TMeta (src/triangle_mesh.jl:1-7): the library keeps one UInt32 per primitive; for a metadata type other
    # than UInt32 that word is an index into this table and the wrapper hands out Triangle{TMeta}(..., meta_table[word]).
    meta_table::Vector{Any}
    meta_type::DataType
    function MI355XTLAS(backend::MI355XBackend = MI355XBackend())
        ref = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:rc_scene_create, LIB), Cint, (Cint, Ref{Ptr{Cvoid}}), backend.device, ref))   # TLAS(backend), :334-358
        tlas = new(backend, ref[], Triangle{UInt32}[], false, Any[], UInt32)
        finalizer(Raycore.free!, tlas)
        return tlas
    end
end
Base.eltype(t::MI355XTLAS) = Triangle{t.meta_type}                                            # Base.eltype(::TLAS), :2334-2341

"The adapted form (StaticTLAS analogue, src/instanced-bvh.jl:155-168): what gets passed to trace calls."
struct MI355XStaticTLAS <: AbstractAdaptedAccel
    owner::MI355XTLAS
end
Base.eltype(a::MI355XStaticTLAS) = eltype(a.owner)

function Raycore.free!(t::MI355XTLAS)                                   # free!, :383-399
    t.ptr == C_NULL && return nothing
    ccall((:rc_scene_destroy, LIB), Cint, (Ptr{Cvoid},), t.ptr)
    t.ptr = C_NULL
    return nothing
end

# ---- mesh ingestion: the GeometryBasics decomposition stays in Julia exactly as in build_and_append_blas! (:581-590); the
# per-face work (index expansion, degenerate filter, build_triangle, LBVH) happens in the library (rc_add_mesh).
function decomposed(mesh::GeometryBasics.Mesh)
    nmesh = GeometryBasics.expand_faceviews(mesh)
    fs = decompose(TriangleFace{UInt32}, nmesh)
    verts = decompose(Point3f, nmesh)
    norms = Raycore.Normal3f.(decompose_normals(nmesh))
    uvs_raw = GeometryBasics.decompose_uv(nmesh)
    indices = collect(reinterpret(UInt32, fs)) .- UInt32(1)                                     # 0-based at the C boundary
    face_meta = hasproperty(nmesh, :face_meta) ? UInt32.(nmesh.face_meta) : nothing             # per vertex after expand_faceviews (:595)
    return (verts = collect(reinterpret(Float32, verts)), normals = collect(reinterpret(Float32, norms)),
            uvs = isnothing(uvs_raw) ? nothing : collect(reinterpret(Float32, Point2f.(uvs_raw))),
            nv = length(verts), indices = indices, nf = length(fs), face_meta = face_meta)
end
function add_mesh!(t, mesh::GeometryBasics.Mesh)
    d = decomposed(mesh)
    blas = Ref{UInt32}(0)
    check(ccall((:rc_add_mesh, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, UInt32, Ptr{UInt32}, UInt32, Ptr{UInt32}, Ref{UInt32}),
                t.ptr, d.verts, d.normals, d.uvs === nothing ? C_NULL : d.uvs, d.nv, d.indices, d.nf,
                d.face_meta === nothing ? C_NULL : d.face_meta, blas))
    return blas[]
end

xforms_buffer(ts::AbstractVector{Mat3x4f}) = collect(reinterpret(Float32, ts))     # Mat3x4f bytes == Vulkan 3x4 (:28-31)
xforms_buffer(ts::AbstractVector) = xforms_buffer(map(mat4_to_mat3x4, ts))

function Base.push!(t::MI355XTLAS, mesh::GeometryBasics.Mesh, transforms::AbstractVector;
                    instance_ids::Union{Nothing, AbstractVector{<:Integer}} = nothing, sbt_offset::UInt32 = UInt32(0))
    instance_ids !== nothing && length(instance_ids) != length(transforms) &&
        throw(ArgumentError("instance_ids length $(length(instance_ids)) != transforms length $(length(transforms))"))   # :664-666
    blas = Ref{UInt32}(add_mesh!(t, mesh)); handle = Ref{UInt32}(0)
    ids = instance_ids === nothing ? C_NULL : UInt32.(instance_ids)
    check(ccall((:rc_add_instances, LIB), Cint, (Ptr{Cvoid}, UInt32, Ptr{Float32}, Ptr{UInt32}, UInt32, Ref{UInt32}),
                t.ptr, blas[], xforms_buffer(transforms), ids, length(transforms), handle))
    t.prims_valid = false
    return TLASHandle(handle[])
end
Base.push!(t::MI355XTLAS, mesh::GeometryBasics.Mesh, transform = Raycore.Mat4f(Raycore.I);
           instance_id::UInt32 = UInt32(0), sbt_offset::UInt32 = UInt32(0)) =
    push!(t, mesh, [transform]; instance_ids = [instance_id])                                   # :639-646

function Base.delete!(t::MI355XTLAS, h::TLASHandle)::Bool                                       # :690-699
    d = Ref{Cint}(0)
    check(ccall((:rc_delete, LIB), Cint, (Ptr{Cvoid}, UInt32, Ref{Cint}), t.ptr, h.id, d))
    return d[] != 0
end

function Raycore.update_transforms!(t::MI355XTLAS, h::TLASHandle, transforms::AbstractVector)  # :784-797
    check(ccall((:rc_update_transforms, LIB), Cint, (Ptr{Cvoid}, UInt32, Ptr{Float32}, UInt32),
                t.ptr, h.id, xforms_buffer(transforms), length(transforms)))
    return nothing
end
function Raycore.update_transform!(t::MI355XTLAS, h::TLASHandle, transform)                    # :755-770
    n = Raycore.n_instances(t, h)
    Raycore.is_valid(t, h) && n != 1 && error("Handle has $n instances, use update_transforms! for multiple")
    Raycore.update_transforms!(t, h, [transform])
end
function Raycore.update!(t::MI355XTLAS, h::TLASHandle, mesh::GeometryBasics.Mesh)              # :808-857
    d = decomposed(mesh)
    check(ccall((:rc_update_geometry_mesh, LIB), Cint, (Ptr{Cvoid}, UInt32, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, UInt32, Ptr{UInt32}, UInt32, Ptr{UInt32}),
                t.ptr, h.id, d.verts, d.normals, d.uvs === nothing ? C_NULL : d.uvs, d.nv, d.indices, d.nf,
                d.face_meta === nothing ? C_NULL : d.face_meta))
    t.prims_valid = false
    return nothing
end

function Raycore.sync!(t::MI355XTLAS)                                                         # :894-921
    action = Ref{Cint}(0)
    check(ccall((:rc_sync, LIB), Cint, (Ptr{Cvoid}, Ref{Cint}), t.ptr, action))
    action[] == 2 && (t.prims_valid = false)
    return t
end

# Adapt.adapt(backend, tlas): sync, then hand out the adapted form; cross-backend adapt errors loudly (:1085-1102)
function Adapt.adapt_structure(to, t::MI355XTLAS)
    to isa MI355XBackend || to === nothing ||
        error("Cross-backend Adapt.adapt(::$(typeof(to)), ::MI355XTLAS) is not supported.")
    Raycore.sync!(t)
    return MI355XStaticTLAS(t)
end

function Raycore.is_valid(t::MI355XTLAS, h::TLASHandle)::Bool                                  # :524-526
    v = Ref{Cint}(0); check(ccall((:rc_is_valid, LIB), Cint, (Ptr{Cvoid}, UInt32, Ref{Cint}), t.ptr, h.id, v)); v[] != 0
end
function Raycore.n_instances(t::MI355XTLAS, h::TLASHandle)::Int                                # :533-537
    n = Ref{UInt32}(0); check(ccall((:rc_handle_instance_count, LIB), Cint, (Ptr{Cvoid}, UInt32, Ref{UInt32}), t.ptr, h.id, n)); Int(n[])
end
function counts(t::MI355XTLAS)
    c = [Ref{UInt32}(0) for _ in 1:6]
    check(ccall((:rc_counts, LIB), Cint, (Ptr{Cvoid}, Ref{UInt32}, Ref{UInt32}, Ref{UInt32}, Ref{UInt32}, Ref{UInt32}, Ref{UInt32}), t.ptr, c...))
    return map(x -> Int(x[]), c)
end
Raycore.n_instances(t::MI355XTLAS) = counts(t)[1]                                              # :2391-2398
Raycore.n_total_instances(t::MI355XTLAS) = counts(t)[2]
Raycore.n_geometries(t::MI355XTLAS) = counts(t)[3]                                             # :2405
function Raycore.get_instances(t::MI355XTLAS, h::TLASHandle)                                   # :732-738
    n = Ref{UInt32}(0)
    check(ccall((:rc_get_instances, LIB), Cint, (Ptr{Cvoid}, UInt32, Ptr{Cvoid}, UInt32, Ref{UInt32}), t.ptr, h.id, C_NULL, 0, n))
    out = Vector{InstanceDescriptor}(undef, n[])                                               # 108-byte isbits struct, same layout
    check(ccall((:rc_get_instances, LIB), Cint, (Ptr{Cvoid}, UInt32, Ptr{Cvoid}, UInt32, Ref{UInt32}), t.ptr, h.id, out, n[], n))
    return out
end
Raycore.get_instance(t::MI355XTLAS, h::TLASHandle, i::Integer = 1) = Raycore.get_instances(t, h)[i]
function Raycore.world_bound(t::Union{MI355XTLAS, MI355XStaticTLAS})                           # :2147-2149
    o = t isa MI355XTLAS ? t : t.owner
    b = Vector{Float32}(undef, 6); check(ccall((:rc_world_bound, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}), o.ptr, b))
    return Bounds3(Point3f(b[1:3]...), Point3f(b[4:6]...))
end
Raycore.wait_for_gpu!(t::MI355XTLAS) = (check(ccall((:rc_wait, LIB), Cint, (Ptr{Cvoid},), t.ptr)); t)   # :2418-2421

# ---- tracing ---------------------------------------------------------------------------------------------
to_rtray(r::Raycore.Ray) = RTRay(r.o[1], r.o[2], r.o[3], r.t_min, r.d[1], r.d[2], r.d[3], r.t_max)     # src/rt_transport.jl:10-19

function trace(a::MI355XStaticTLAS, rays::Vector{RTRay}; any::Bool = false)
    hits = Vector{RTHitResult}(undef, length(rays))
    f = any ? :rc_trace_any : :rc_trace_closest
    check(ccall((f, LIB), Cint, (Ptr{Cvoid}, Ptr{RTRay}, Ptr{RTHitResult}, UInt64), a.owner.ptr, rays, hits, length(rays)))
    return hits
end

"One host batch on several devices (SURVEY.md 8e: replicas of the scene, contiguous ray shards, no collective): `accels[g]` is the adapted
accel of the same scene on device g; shard g travels over device g's own PCIe link."
function trace(accels::Vector{MI355XStaticTLAS}, rays::Vector{RTRay}; any::Bool = false)
    hits = Vector{RTHitResult}(undef, length(rays))
    ptrs = Ptr{Cvoid}[a.owner.ptr for a in accels]
    f = any ? :rc_trace_any_multi : :rc_trace_closest_multi
    GC.@preserve accels check(ccall((f, LIB), Cint, (Ptr{Ptr{Cvoid}}, Cint, Ptr{RTRay}, Ptr{RTHitResult}, UInt64), ptrs, length(ptrs), rays, hits, length(rays)))
    return hits
end
Raycore.trace_rays(accels::Vector{MI355XStaticTLAS}, rays::AbstractVector{<:Raycore.AbstractRay}) =
    map(h -> result_tuple(accels[1], h, empty_triangle(Triangle{UInt32})), trace(accels, map(to_rtray, rays)))

function primitives(a::MI355XStaticTLAS)            # all_blas_prims as Triangle{UInt32} values: the library's 136-byte records ARE that struct
    t = a.owner
    if !t.prims_valid
        n = Ref{UInt32}(0)
        check(ccall((:rc_export_triangles, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, UInt32, Ref{UInt32}), t.ptr, C_NULL, 0, n))
        t.prims = Vector{Triangle{UInt32}}(undef, n[])       # sizeof(Triangle{UInt32}) == 136 (src/triangle_mesh.jl:1-7)
        check(ccall((:rc_export_triangles, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, UInt32, Ref{UInt32}), t.ptr, t.prims, n[], n))
        t.prims_valid = true
    end
    return t.prims
end

# the 5-tuple of closest_hit / any_hit (:2010-2023, :2106-2139); instance index is 1-based, 0 on a miss
function result_tuple(a::MI355XStaticTLAS, h::RTHitResult, miss_prim)
    h.hit == 0 && return (false, miss_prim, 0f0, SVector{3, Float32}(0, 0, 0), UInt32(0))
    w = 1f0 - h.bary_u - h.bary_v
    return (true, typed_triangle(a.owner, primitives(a)[h.primitive_id + 1]), h.t, SVector{3, Float32}(w, h.bary_u, h.bary_v), h.instance_id + UInt32(1))
end
Raycore.closest_hit(a::MI355XStaticTLAS, ray::Raycore.AbstractRay) =
    result_tuple(a, trace(a, [to_rtray(ray)])[1], empty_triangle(Triangle{UInt32}))
Raycore.any_hit(a::MI355XStaticTLAS, ray::Raycore.AbstractRay) =
    result_tuple(a, trace(a, [to_rtray(ray)]; any = true)[1], primitives(a)[1])                 # dummy = all_blas_prims[1], :2137
Raycore.trace_rays(a::MI355XStaticTLAS, rays::AbstractVector{<:Raycore.AbstractRay}) =         # ext/RaycoreMakieExt.jl:81-87
    map(h -> result_tuple(a, h, empty_triangle(Triangle{UInt32})), trace(a, map(to_rtray, rays)))

# ---- drivers (src/kernels.jl:74-124) ---------------------------------------------------------------------
function Raycore.get_illumination(a::MI355XStaticTLAS, viewdir; grid_size = 1000)
    out = Vector{Float32}(undef, counts(a.owner)[4])
    check(ccall((:rc_get_illumination, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, UInt32, Ptr{Float32}),
                a.owner.ptr, Float32[viewdir...], grid_size, out))
    return out
end
function Raycore.get_illumination(accels::Vector{MI355XStaticTLAS}, viewdir; grid_size = 1000)   # the ray grid in one share per device
    out = Vector{Float32}(undef, counts(accels[1].owner)[4])
    ptrs = Ptr{Cvoid}[a.owner.ptr for a in accels]
    GC.@preserve accels check(ccall((:rc_get_illumination_multi, LIB), Cint, (Ptr{Ptr{Cvoid}}, Cint, Ptr{Float32}, UInt32, Ptr{Float32}),
                                    ptrs, length(ptrs), Float32[viewdir...], grid_size, out))
    return out
end
function Raycore.view_factors(a::MI355XStaticTLAS; rays_per_triangle = 10000, seed::UInt64 = rand(UInt64))
    n = counts(a.owner)[4]
    out = Matrix{UInt32}(undef, n, n)      # column-major, [src_meta, hit_meta] as in the reference
    check(ccall((:rc_view_factors, LIB), Cint, (Ptr{Cvoid}, UInt32, UInt64, Ptr{UInt32}), a.owner.ptr, rays_per_triangle, seed, out))
    return out
end
"""
    view_factors(accels::Vector{MI355XStaticTLAS}; rays_per_triangle, seed, mode = :rows) -> Matrix{UInt32}

view_factors sharded over the GPUs of one node (SURVEY.md 8e): `accels[g]` is the adapted accel of a TLAS built from the same meshes on
device g.  `mode = :rows`: accel g traces matrix rows [gN/G, (g+1)N/G) and copies its chunks straight into the result over its own PCIe
link (no xGMI traffic, no collective); `mode = :rays`: every device shoots rays_per_triangle / G rays of every source and the row chunks
are summed on accels[1]'s device by RCCL `ncclReduce` over xGMI while the next chunk is traced.  Same matrix as the single-device call,
bit for bit (Philox keyed by ray index and source primitive).
"""
function Raycore.view_factors(accels::Vector{MI355XStaticTLAS}; rays_per_triangle = 10000, seed::UInt64 = rand(UInt64), mode::Symbol = :rows)
    n = counts(accels[1].owner)[4]
    out = Matrix{UInt32}(undef, n, n)
    ptrs = Ptr{Cvoid}[a.owner.ptr for a in accels]
    GC.@preserve accels check(ccall((:rc_view_factors_multi, LIB), Cint, (Ptr{Ptr{Cvoid}}, Cint, UInt32, UInt64, Ptr{UInt32}, Cint),
                                    ptrs, length(ptrs), rays_per_triangle, seed, out, mode === :rays ? 1 : 0))
    return out
end
"""
    view_factor_totals(accel_or_accels; rays_per_triangle, seed) -> (received::Vector{UInt64}, emitted::Vector{UInt64})

The per-triangle totals of `view_factors` without the N x N matrix: `received[j] == sum(view(view_factors(...), :, j))` -- the rays that
arrive at the triangles with metadata j, what docs/src/viewfactors_content.md:62-68 computes from the matrix -- and `emitted[i] ==
sum(view(view_factors(...), i, :))`, for the same seed.  With a vector of accels (one per GPU, same meshes) the RAYS are sharded: device g
shoots ray indices [gR/G, (g+1)R/G) of every source and one RCCL `ncclReduce` (UInt64, 2N elements) over xGMI sums the vectors on
accels[1]'s device.  Costs the tracing (C5: ~40 ms on one MI355X), not 10 GB over PCIe.
"""
function view_factor_totals(a::MI355XStaticTLAS; rays_per_triangle = 10000, seed::UInt64 = rand(UInt64))
    n = counts(a.owner)[4]
    received, emitted = Vector{UInt64}(undef, n), Vector{UInt64}(undef, n)
    check(ccall((:rc_view_factor_totals, LIB), Cint, (Ptr{Cvoid}, UInt32, UInt64, Ptr{UInt64}, Ptr{UInt64}), a.owner.ptr, rays_per_triangle, seed, received, emitted))
    return received, emitted
end
function view_factor_totals(accels::Vector{MI355XStaticTLAS}; rays_per_triangle = 10000, seed::UInt64 = rand(UInt64))
    n = counts(accels[1].owner)[4]
    received, emitted = Vector{UInt64}(undef, n), Vector{UInt64}(undef, n)
    ptrs = Ptr{Cvoid}[a.owner.ptr for a in accels]
    GC.@preserve accels check(ccall((:rc_view_factor_totals_multi, LIB), Cint, (Ptr{Ptr{Cvoid}}, Cint, UInt32, UInt64, Ptr{UInt64}, Ptr{UInt64}),
                                    ptrs, length(ptrs), rays_per_triangle, seed, received, emitted))
    return received, emitted
end

# Once per set of devices, before the *_multi calls that are timed: RCCL + communicator, auxiliary streams, staging vectors, a warm-up
# collective (rc_multi_prepare).  Returns (comm_init_ms, streams_and_buffers_ms, warmup_collective_ms, total_ms, rccl_ranks).
function multi_prepare(accels::Vector{MI355XStaticTLAS})
    handles = Ptr{Cvoid}[a.owner.ptr for a in accels]
    ms = zeros(Float32, 4)
    ranks = Ref{Cint}(0)
    GC.@preserve accels begin
        check(ccall((:rc_multi_prepare, LIB), Cint, (Ptr{Ptr{Cvoid}}, Cint, Ptr{Float32}), handles, length(handles), ms))
        check(ccall((:rc_multi_ranks, LIB), Cint, (Ptr{Ptr{Cvoid}}, Cint, Ptr{Cint}), handles, length(handles), ranks))
    end
    return (comm_init_ms = ms[1], streams_and_buffers_ms = ms[2], warmup_collective_ms = ms[3], total_ms = ms[4], rccl_ranks = Int(ranks[]))
end

"One shard of the totals (sources [src_begin, src_end) x rays [ray_begin, ray_end)), ACCUMULATED into device vectors of N UInt64 each: the
unit of a multi-process run (MPI.jl / one process per GPU: reduce 2N UInt64)."
view_factor_totals_device!(a::MI355XStaticTLAS, rays_per_triangle::Integer, seed::UInt64, src_begin::Integer, src_end::Integer, ray_begin::Integer,
                           ray_end::Integer, d_received::Ptr{UInt64}, d_emitted::Ptr{UInt64}; stream::Ptr{Cvoid} = C_NULL) =
    check(ccall((:rc_view_factor_totals_device, LIB), Cint, (Ptr{Cvoid}, UInt32, UInt64, UInt32, UInt32, UInt32, UInt32, Ptr{UInt64}, Ptr{UInt64}, Ptr{Cvoid}),
                a.owner.ptr, rays_per_triangle, seed, src_begin, src_end, ray_begin, ray_end, d_received, d_emitted, stream))
"Rows [row_begin, row_end) (0-based, end exclusive) of the matrix into `out` (column-major, any leading dimension >= N): the unit of a
multi-process run, where every process maps the same shared-memory matrix (`Mmap.mmap` of a file in /dev/shm) and fills its own rows."
function view_factors_rows!(out::AbstractMatrix{UInt32}, a::MI355XStaticTLAS, row_begin::Integer, row_end::Integer; rays_per_triangle = 10000, seed::UInt64)
    check(ccall((:rc_view_factors_rows_host, LIB), Cint, (Ptr{Cvoid}, UInt32, UInt64, UInt32, UInt32, Ptr{UInt32}, UInt64),
                a.owner.ptr, rays_per_triangle, seed, row_begin, row_end, out, stride(out, 2)))
    return out
end

# the reference's RayHit (src/kernels.jl:1-5)
struct RayHit{T}
    hit::Bool
    point::Point3f
    metadata::T
end
metadata_of(t::MI355XTLAS, word::UInt32) = t.meta_type === UInt32 ? word : t.meta_table[word]
typed_triangle(t::MI355XTLAS, tri::Triangle{UInt32}) = t.meta_type === UInt32 ? tri :
    Triangle(tri.vertices, tri.normals, tri.tangents, tri.uv, metadata_of(t, tri.metadata))

"hits_from_grid (src/kernels.jl:58-72): the ray grid is generated on the device (rc_generate_ray_grid_device is what the fused
 drivers use; here the host-buffer trace is enough), hit point = sum_mul(bary, prim.vertices) in Float32."
function hits_from_grid(a::MI355XStaticTLAS, viewdir; grid_size = 32)
    dir = normalize(Vec3f(viewdir))
    rays = ray_grid(a, dir, grid_size)
    hits = trace(a, rays)
    prims = primitives(a)
    T = a.owner.meta_type
    out = Matrix{RayHit{T}}(undef, grid_size, grid_size)
    for k in eachindex(hits)
        h = hits[k]
        if h.hit == 0
            out[k] = RayHit{T}(false, Point3f(0), metadata_of(a.owner, empty_triangle(Triangle{UInt32}).metadata + UInt32(T === UInt32 ? 0 : 1)))
        else
            p = prims[h.primitive_id + 1]
            w = (1f0 - h.bary_u) - h.bary_v
            out[k] = RayHit{T}(true, Point3f(w * p.vertices[1] + h.bary_u * p.vertices[2] + h.bary_v * p.vertices[3]), metadata_of(a.owner, p.metadata))
        end
    end
    return out
end
"get_centroid (src/kernels.jl:106-110)"
function Raycore.get_centroid(a::MI355XStaticTLAS, viewdir; grid_size = 32)
    hits = hits_from_grid(a, viewdir; grid_size = grid_size)
    pts = [h.point for h in hits if h.hit]
    return pts, sum(pts) / length(pts)
end

# ---- device-pointer entry points: rays / hits / matrices that already live in HBM (ROCArray pointers, a HIP stream) ---------------
"generate_ray_grid (src/kernels.jl:10-56) into a device buffer of grid^2 RTRay records."
ray_grid_device!(a::MI355XStaticTLAS, viewdir, grid_size::Integer, d_rays::Ptr{RTRay}; stream::Ptr{Cvoid} = C_NULL) =
    check(ccall((:rc_generate_ray_grid_device, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, UInt32, Ptr{RTRay}, Ptr{Cvoid}),
                a.owner.ptr, Float32[viewdir...], grid_size, d_rays, stream))
function ray_grid(a::MI355XStaticTLAS, viewdir, grid_size::Integer)   # host copy, through the host-buffer illumination path's generator
    n = grid_size * grid_size
    rays = Vector{RTRay}(undef, n)
    # a device scratch buffer is the caller's business in a GPU pipeline; the host form stages through AMDGPU.jl when it is loaded
    buf = device_alloc(sizeof(RTRay) * n)
    ray_grid_device!(a, viewdir, grid_size, Ptr{RTRay}(buf))
    Raycore.wait_for_gpu!(a.owner)
    device_download!(rays, buf)
    device_free(buf)
    return rays
end
# minimal device-memory helpers over the HIP runtime the library already links (hipMalloc / hipMemcpy / hipFree)
const HIP = get(ENV, "RAYCORE_MI355X_HIP", "libamdhip64.so")
function device_alloc(bytes::Integer)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    ccall((:hipMalloc, HIP), Cint, (Ref{Ptr{Cvoid}}, Csize_t), p, bytes) == 0 || error("hipMalloc failed")
    return p[]
end
device_free(p::Ptr{Cvoid}) = (ccall((:hipFree, HIP), Cint, (Ptr{Cvoid},), p); nothing)
device_download!(dst::Vector, src::Ptr{Cvoid}) =
    (ccall((:hipMemcpy, HIP), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, Csize_t, Cint), dst, src, sizeof(dst), 2) == 0 || error("hipMemcpy failed"); dst)

"closest_hit / any_hit over device-resident RTRay / RTHitResult arrays, asynchronous on `stream` (errors surface at wait_for_gpu!)."
trace_device!(a::MI355XStaticTLAS, d_rays::Ptr{RTRay}, d_hits::Ptr{RTHitResult}, n::Integer; any::Bool = false, stream::Ptr{Cvoid} = C_NULL) =
    check(any ? ccall((:rc_trace_any_device, LIB), Cint, (Ptr{Cvoid}, Ptr{RTRay}, Ptr{RTHitResult}, UInt64, Ptr{Cvoid}), a.owner.ptr, d_rays, d_hits, n, stream) :
                ccall((:rc_trace_closest_device, LIB), Cint, (Ptr{Cvoid}, Ptr{RTRay}, Ptr{RTHitResult}, UInt64, Ptr{Cvoid}), a.owner.ptr, d_rays, d_hits, n, stream))

"Several INDEPENDENT device batches in one call: they overlap on the scene's auxiliary streams, forked from and joined back into `stream`."
function trace_device_batches!(a::MI355XStaticTLAS, d_rays::Vector{Ptr{RTRay}}, d_hits::Vector{Ptr{RTHitResult}}, n::Vector{UInt64}; any::Bool = false, stream::Ptr{Cvoid} = C_NULL)
    length(d_rays) == length(d_hits) == length(n) || error("d_rays, d_hits and n must have one entry per batch")
    check(any ? ccall((:rc_trace_any_device_batches, LIB), Cint, (Ptr{Cvoid}, Ptr{Ptr{RTRay}}, Ptr{Ptr{RTHitResult}}, Ptr{UInt64}, Cint, Ptr{Cvoid}), a.owner.ptr, d_rays, d_hits, n, length(n), stream) :
                ccall((:rc_trace_closest_device_batches, LIB), Cint, (Ptr{Cvoid}, Ptr{Ptr{RTRay}}, Ptr{Ptr{RTHitResult}}, Ptr{UInt64}, Cint, Ptr{Cvoid}), a.owner.ptr, d_rays, d_hits, n, length(n), stream))
end

"One shard of get_illumination: rays [ray_begin, ray_end) of the grid, histogram ACCUMULATED into the device vector d_counts."
illumination_device!(a::MI355XStaticTLAS, viewdir, grid_size::Integer, ray_begin::Integer, ray_end::Integer, d_counts::Ptr{Float32};
                     stream::Ptr{Cvoid} = C_NULL) =
    check(ccall((:rc_get_illumination_device, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, UInt32, UInt64, UInt64, Ptr{Float32}, Ptr{Cvoid}),
                a.owner.ptr, Float32[viewdir...], grid_size, ray_begin, ray_end, d_counts, stream))

"One shard of view_factors! (src/kernels.jl:80-104): source primitives [src_begin, src_end) x rays [ray_begin, ray_end), counts
 ACCUMULATED into d_matrix at row * row_stride + col * col_stride (row = src_meta - 1, or the source's sorted position - row_offset
 when by_primitive); the multi-GPU drivers (one process per GPU, RCCL) are built on this entry point."
view_factors_device!(a::MI355XStaticTLAS, rays_per_triangle::Integer, seed::UInt64, src_begin::Integer, src_end::Integer,
                     ray_begin::Integer, ray_end::Integer, d_matrix::Ptr{UInt32}, row_stride::Integer, col_stride::Integer;
                     row_offset::Integer = 0, by_primitive::Bool = false, stream::Ptr{Cvoid} = C_NULL) =
    check(ccall((:rc_view_factors_device, LIB), Cint,
                (Ptr{Cvoid}, UInt32, UInt64, UInt32, UInt32, UInt32, UInt32, Ptr{UInt32}, UInt64, UInt64, UInt32, UInt32, Ptr{Cvoid}),
                a.owner.ptr, rays_per_triangle, seed, src_begin, src_end, ray_begin, ray_end, d_matrix, row_stride, col_stride,
                row_offset, by_primitive ? 1 : 0, stream))
"The rays view_factors shoots for one source primitive (for inspection / tests)."
view_factor_rays_device!(a::MI355XStaticTLAS, seed::UInt64, src_prim::Integer, ray_begin::Integer, n_rays::Integer, d_rays::Ptr{RTRay};
                         stream::Ptr{Cvoid} = C_NULL) =
    check(ccall((:rc_view_factor_rays_device, LIB), Cint, (Ptr{Cvoid}, UInt64, UInt32, UInt32, UInt32, Ptr{RTRay}, Ptr{Cvoid}),
                a.owner.ptr, seed, src_prim, ray_begin, n_rays, d_rays, stream))

# wavefront stages next to the trace (docs/src/wavefront-renderer.jl:219-476)
hit_points_device!(a::MI355XStaticTLAS, d_rays::Ptr{RTRay}, d_hits::Ptr{RTHitResult}, n::Integer, d_points::Ptr{Float32},
                   d_normals::Ptr{Float32} = Ptr{Float32}(C_NULL); stream::Ptr{Cvoid} = C_NULL) =
    check(ccall((:rc_hit_points_device, LIB), Cint, (Ptr{Cvoid}, Ptr{RTRay}, Ptr{RTHitResult}, UInt64, Ptr{Float32}, Ptr{Float32}, Ptr{Cvoid}),
                a.owner.ptr, d_rays, d_hits, n, d_points, d_normals, stream))
shadow_rays_device!(a::MI355XStaticTLAS, d_rays::Ptr{RTRay}, d_hits::Ptr{RTHitResult}, n::Integer, light, d_out::Ptr{RTRay};
                    bias::Float32 = 0.01f0, stream::Ptr{Cvoid} = C_NULL) =
    check(ccall((:rc_shadow_rays_device, LIB), Cint, (Ptr{Cvoid}, Ptr{RTRay}, Ptr{RTHitResult}, UInt64, Ptr{Float32}, Cfloat, Ptr{RTRay}, Ptr{Cvoid}),
                a.owner.ptr, d_rays, d_hits, n, Float32[light...], bias, d_out, stream))
reflection_rays_device!(a::MI355XStaticTLAS, d_rays::Ptr{RTRay}, d_hits::Ptr{RTHitResult}, n::Integer, d_out::Ptr{RTRay};
                        bias::Float32 = 0.01f0, stream::Ptr{Cvoid} = C_NULL) =
    check(ccall((:rc_reflection_rays_device, LIB), Cint, (Ptr{Cvoid}, Ptr{RTRay}, Ptr{RTHitResult}, UInt64, Cfloat, Ptr{RTRay}, Ptr{Cvoid}),
                a.owner.ptr, d_rays, d_hits, n, bias, d_out, stream))
shading_attributes_device!(a::MI355XStaticTLAS, d_hits::Ptr{RTHitResult}, n::Integer, d_normals::Ptr{Float32}, d_uvs::Ptr{Float32};
                           stream::Ptr{Cvoid} = C_NULL) =
    check(ccall((:rc_shading_attributes_device, LIB), Cint, (Ptr{Cvoid}, Ptr{RTHitResult}, UInt64, Ptr{Float32}, Ptr{Float32}, Ptr{Cvoid}),
                a.owner.ptr, d_hits, n, d_normals, d_uvs, stream))
compact_hits_device!(a::MI355XStaticTLAS, d_hits::Ptr{RTHitResult}, n::Integer, d_indices::Ptr{UInt32}, d_count::Ptr{UInt32};
                     stream::Ptr{Cvoid} = C_NULL) =
    check(ccall((:rc_compact_hits_device, LIB), Cint, (Ptr{Cvoid}, Ptr{RTHitResult}, UInt64, Ptr{UInt32}, Ptr{UInt32}, Ptr{Cvoid}),
                a.owner.ptr, d_hits, n, d_indices, d_count, stream))
primary_rays_lookat_device!(a::MI355XStaticTLAS, pos, right, up, forward, half_width::Float32, half_height::Float32, width::Integer,
                            height::Integer, d_rays::Ptr{RTRay}; samples::Integer = 1, seed::UInt64 = UInt64(0), jitter::Bool = true,
                            stream::Ptr{Cvoid} = C_NULL) =
    check(ccall((:rc_primary_rays_lookat_device, LIB), Cint,
                (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, Cfloat, Cfloat, UInt32, UInt32, UInt32, UInt64, Cint, Ptr{RTRay}, Ptr{Cvoid}),
                a.owner.ptr, Float32[pos...], Float32[right...], Float32[up...], Float32[forward...], half_width, half_height, width, height,
                samples, seed, jitter ? 1 : 0, d_rays, stream))

# instance_buffer(tlas, handle) + refit_tlas!(tlas) (src/Raycore.jl:117-128, src/instanced-bvh.jl:2197-2222): rewrite the handle's
# InstanceDescriptor records on the device (the caller's own kernel), then commit without a host round trip.
function Raycore.instance_buffer(t::MI355XTLAS, h::TLASHandle)
    p = Ref{Ptr{InstanceDescriptor}}(C_NULL); n = Ref{UInt32}(0)
    check(ccall((:rc_instance_buffer_device, LIB), Cint, (Ptr{Cvoid}, UInt32, Ref{Ptr{InstanceDescriptor}}, Ref{UInt32}), t.ptr, h.id, p, n))
    return p[], Int(n[])
end
Raycore.refit_tlas!(t::MI355XTLAS; recompute_inverse::Bool = true) =
    (check(ccall((:rc_refit_device, LIB), Cint, (Ptr{Cvoid}, Cint), t.ptr, recompute_inverse ? 1 : 0)); t)

# collide_instances / collide_instances_any (src/collision.jl:189-262)
function Raycore.collide_instances(t::MI355XTLAS)
    n = Ref{UInt64}(0)
    check(ccall((:rc_collide_instances, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, UInt64, Ref{UInt64}), t.ptr, C_NULL, 0, n))
    out = Vector{Raycore.ContactPair}(undef, n[])                                              # 2 x UInt32, same layout as rc_contact_pair
    n[] > 0 && check(ccall((:rc_collide_instances, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, UInt64, Ref{UInt64}), t.ptr, out, n[], n))
    return out
end
function Raycore.collide_instances_any(t::MI355XTLAS, a::TLASHandle, b::TLASHandle)::Bool
    o = Ref{Cint}(0)
    check(ccall((:rc_collide_instances_any, LIB), Cint, (Ptr{Cvoid}, UInt32, UInt32, Ref{Cint}), t.ptr, a.id, b.id, o))
    return o[] != 0
end

collide_instances_device!(t::MI355XTLAS, d_out::Ptr{Raycore.ContactPair}, capacity::Integer; stream::Ptr{Cvoid} = C_NULL) = begin
    n = Ref{UInt64}(0)
    check(ccall((:rc_collide_instances_device, LIB), Cint, (Ptr{Cvoid}, Ptr{Raycore.ContactPair}, UInt64, Ref{UInt64}, Ptr{Cvoid}), t.ptr, d_out, capacity, n, stream))
    Int(n[])
end

# ---- triangle soup without a GeometryBasics mesh: 9 Float32 per triangle (v0 v1 v2) + one UInt32 of metadata each -----------------
"build_blas(triangles) + push! of its instances (src/instanced-bvh.jl:1376-1443, 661-684) for a caller that already holds plain triangles."
function push_triangles!(t::MI355XTLAS, verts::AbstractMatrix{Float32}, meta::Union{Nothing, Vector{UInt32}}, transforms::AbstractVector;
                         instance_ids::Union{Nothing, AbstractVector{<:Integer}} = nothing)
    size(verts, 1) == 9 || throw(ArgumentError("verts must be 9 x n (column-major: one triangle per column)"))
    blas = Ref{UInt32}(0); handle = Ref{UInt32}(0)
    check(ccall((:rc_add_blas, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{UInt32}, UInt32, Ref{UInt32}),
                t.ptr, verts, meta === nothing ? C_NULL : meta, size(verts, 2), blas))
    check(ccall((:rc_add_instances, LIB), Cint, (Ptr{Cvoid}, UInt32, Ptr{Float32}, Ptr{UInt32}, UInt32, Ref{UInt32}),
                t.ptr, blas[], xforms_buffer(transforms), instance_ids === nothing ? C_NULL : UInt32.(instance_ids), length(transforms), handle))
    t.prims_valid = false
    return TLASHandle(handle[])
end
"The same with the soup already in HBM (e.g. written by a simulation kernel): no PCIe traffic, the LBVH is built where the data is."
function add_triangles_device!(t::MI355XTLAS, d_verts::Ptr{Float32}, d_meta::Ptr{UInt32}, n::Integer)
    blas = Ref{UInt32}(0)
    check(ccall((:rc_add_blas_device, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{UInt32}, UInt32, Ref{UInt32}), t.ptr, d_verts, d_meta, n, blas))
    t.prims_valid = false
    return blas[]
end
"update!(tlas, handle, triangles) for plain triangles (:808-857)."
function update_triangles!(t::MI355XTLAS, h::TLASHandle, verts::AbstractMatrix{Float32}, meta::Union{Nothing, Vector{UInt32}} = nothing)
    check(ccall((:rc_update_geometry, LIB), Cint, (Ptr{Cvoid}, UInt32, Ptr{Float32}, Ptr{UInt32}, UInt32),
                t.ptr, h.id, verts, meta === nothing ? C_NULL : meta, size(verts, 2)))
    t.prims_valid = false
    return nothing
end

# ---- the StaticTLAS fields, read back in the reference's own layouts (src/instanced-bvh.jl:155-168) -------------------------------
function export_array(a::MI355XStaticTLAS, sym::Symbol, ::Type{T}) where {T}
    n = Ref{UInt32}(0)
    export_call(sym, a.owner.ptr, C_NULL, UInt32(0), n)
    out = Vector{T}(undef, n[])
    n[] > 0 && export_call(sym, a.owner.ptr, pointer(out), n[], n)
    return out
end
function export_call(sym::Symbol, scene, out, cap, n)
    sym === :nodes ? check(ccall((:rc_export_tlas_nodes, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, UInt32, Ref{UInt32}), scene, out, cap, n)) :
    sym === :all_blas_nodes ? check(ccall((:rc_export_blas_nodes, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, UInt32, Ref{UInt32}), scene, out, cap, n)) :
    sym === :instances ? check(ccall((:rc_export_instances, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, UInt32, Ref{UInt32}), scene, out, cap, n)) :
    sym === :blas_descriptors ? check(ccall((:rc_export_blas_descs, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, UInt32, Ref{UInt32}), scene, out, cap, n)) :
    sym === :prims ? check(ccall((:rc_export_prims, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, UInt32, Ref{UInt32}), scene, out, cap, n)) :
    error("unknown export $sym")
end
struct PrimRecord            # 40 bytes: what the traversal keeps of a triangle (vertices + metadata word)
    v::NTuple{9, Float32}
    meta::UInt32
end
function Base.getproperty(a::MI355XStaticTLAS, f::Symbol)
    f === :nodes ? export_array(a, :nodes, Raycore.BVHNode2) :                                  # 60-byte BVHNode2 records
    f === :all_blas_nodes ? export_array(a, :all_blas_nodes, Raycore.BVHNode2) :
    f === :instances ? export_array(a, :instances, InstanceDescriptor) :
    f === :blas_descriptors ? export_array(a, :blas_descriptors, Raycore.BLASDescriptor) :
    f === :all_blas_prims ? map(p -> typed_triangle(getfield(a, :owner), p), primitives(a)) :
    f === :prim_records ? export_array(a, :prims, PrimRecord) :
    f === :root_aabb ? Raycore.world_bound(a) :
    getfield(a, f)
end

# scene files (no counterpart in the reference, SURVEY section 5) and the library's tuning options
save(t::MI355XTLAS, path::AbstractString) = check(ccall((:rc_scene_save, LIB), Cint, (Ptr{Cvoid}, Cstring), t.ptr, path))
function load(backend::MI355XBackend, path::AbstractString)
    ref = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:rc_scene_load, LIB), Cint, (Cint, Cstring, Ref{Ptr{Cvoid}}), backend.device, path, ref))
    t = MI355XTLAS(backend)            # a fresh handle whose scene is replaced by the loaded one
    Raycore.free!(t); t.ptr = ref[]
    return t
end
set_option!(t::MI355XTLAS, name::AbstractString, value::Integer) =
    check(ccall((:rc_set_option, LIB), Cint, (Ptr{Cvoid}, Cstring, Int64), t.ptr, name, value))
function get_option(t::MI355XTLAS, name::AbstractString)
    v = Ref{Int64}(0); check(ccall((:rc_get_option, LIB), Cint, (Ptr{Cvoid}, Cstring, Ref{Int64}), t.ptr, name, v)); v[]
end
function last_kernel_ms(t::MI355XTLAS)
    ms = Ref{Cfloat}(0); check(ccall((:rc_last_kernel_ms, LIB), Cint, (Ptr{Cvoid}, Ref{Cfloat}), t.ptr, ms)); ms[]
end
"Durations (ms, oldest first) of the scene's most recent launches, from the events the launches carried themselves."
function recent_kernel_ms(t::MI355XTLAS, max_launches::Integer=47)
    ms = Vector{Cfloat}(undef, max_launches); n = Ref{UInt32}(0)
    check(ccall((:rc_recent_kernel_ms, LIB), Cint, (Ptr{Cvoid}, UInt32, Ptr{Cfloat}, Ref{UInt32}), t.ptr, UInt32(max_launches), ms, n)); ms[1:n[]]
end
device_count() = Int(ccall((:rc_device_count, LIB), Cint, ()))
"Page-lock a host array that is traced again and again (`Vector{RTRay}`, a reused `Vector{RTHitResult}`): DMA at the full PCIe rate."
host_register!(t::MI355XTLAS, a::Array) = (check(ccall((:rc_host_register, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}, UInt64), t.ptr, a, sizeof(a))); a)
host_unregister!(t::MI355XTLAS, a::Array) = (check(ccall((:rc_host_unregister, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), t.ptr, a)); a)

# ---- the reference's two convenience constructors ---------------------------------------------------------------------------------
"""
    MI355XTLAS(items, metadata_fn; backend) -> adapted accel          (TLAS(primitives, metadata_fn; backend), src/instanced-bvh.jl:2276-2324)

One BLAS + one identity instance per item, `instance_id = item index`, per-triangle metadata = `metadata_fn(item_idx, face_idx)`
evaluated BEFORE the degenerate filter drops faces (the face index is the mesh's), `TMetadata = typeof(metadata_fn(1, 1))`.
"""
function MI355XTLAS(items::AbstractVector, metadata_fn::Function; backend::MI355XBackend = MI355XBackend())
    t = MI355XTLAS(backend)
    T = typeof(metadata_fn(1, 1))
    t.meta_type = T
    ident = Float32[1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0]
    for (mi, item) in enumerate(items)
        mesh = item isa GeometryBasics.Mesh ? item : GeometryBasics.uv_normal_mesh(item)
        d = decomposed(mesh)
        # ONE WORD PER FACE (rc_add_mesh_face_metadata): faces of a face-view-expanded mesh share vertices -- the two triangles of a quad
        # (a, b, c), (a, c, d) share their first vertex -- so a per-vertex array (what rc_add_mesh reads, the push! path's face_meta)
        # cannot carry metadata_fn(mi, fi); the reference calls it per face (:2300-2306)
        words = Vector{UInt32}(undef, d.nf)
        for fi in 1:d.nf
            m = metadata_fn(mi, fi)
            words[fi] = T === UInt32 ? m : (push!(t.meta_table, m); UInt32(length(t.meta_table)))
        end
        blas = Ref{UInt32}(0); handle = Ref{UInt32}(0)
        check(ccall((:rc_add_mesh_face_metadata, LIB), Cint, (Ptr{Cvoid}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}, UInt32, Ptr{UInt32}, UInt32, Ptr{UInt32}, Ref{UInt32}),
                    t.ptr, d.verts, d.normals, d.uvs === nothing ? C_NULL : d.uvs, d.nv, d.indices, d.nf, words, blas))
        check(ccall((:rc_add_instances_with_inverse, LIB), Cint, (Ptr{Cvoid}, UInt32, Ptr{Float32}, Ptr{Float32}, Ptr{UInt32}, UInt32, Ref{UInt32}),
                    t.ptr, blas[], ident, ident, UInt32[mi], 1, handle))                          # identity for both matrices, :2314-2320
    end
    return Adapt.adapt(backend, t)
end
"""
    MI355XTLAS(meshes; backend) -> (tlas, handles)                      (TLAS(meshes; backend), src/instanced-bvh.jl:2361-2378)
"""
function MI355XTLAS(meshes::AbstractVector{<:GeometryBasics.Mesh}; backend::MI355XBackend = MI355XBackend())
    isempty(meshes) && error("Cannot create TLAS from empty mesh list")
    t = MI355XTLAS(backend)
    handles = TLASHandle[push!(t, m) for m in meshes]
    Raycore.sync!(t)
    return t, handles
end

# ---- BVH4 (src/bvh4.jl): BLAS-level 4-wide tree, collapsed on the device ------------------------------------
struct MI355XBLAS4
    scene::MI355XTLAS            # private scene holding the geometry
    blas_id::UInt32
    num_interior::Int32          # length(nodes), as build_blas4 stores it (:521)
end
function build_blas4_mi355x(backend::MI355XBackend, mesh::GeometryBasics.Mesh)                   # build_blas4, :511-522
    t = MI355XTLAS(backend)
    id = Ref{UInt32}(add_mesh!(t, mesh)); n = Ref{UInt32}(0)
    check(ccall((:rc_blas4_build, LIB), Cint, (Ptr{Cvoid}, UInt32, Ref{UInt32}), t.ptr, id[], n))
    return MI355XBLAS4(t, id[], Int32(n[]))
end
function nodes(b::MI355XBLAS4)                                                                  # Vector{BVHNode4}, 120 B each (:40-69)
    out = Vector{Raycore.BVHNode4}(undef, b.num_interior)
    check(ccall((:rc_export_blas4_nodes, LIB), Cint, (Ptr{Cvoid}, UInt32, Ptr{Cvoid}, UInt32, Ptr{UInt32}), b.scene.ptr, b.blas_id, out, length(out), C_NULL))
    return out
end
function trace4(b::MI355XBLAS4, rays::Vector{RTRay}; any::Bool = false)                         # closest_hit4 :606-689 / any_hit4 :696-766 over a batch
    hits = Vector{RTHitResult}(undef, length(rays))
    f = any ? :rc_trace_any4 : :rc_trace_closest4
    check(ccall((f, LIB), Cint, (Ptr{Cvoid}, UInt32, Ptr{RTRay}, Ptr{RTHitResult}, UInt64), b.scene.ptr, b.blas_id, rays, hits, length(rays)))
    return hits     # (hit, primitives[primitive_id + 1], t, (1 - u - v, u, v)) per ray; no instance index at this level
end
trace4_device!(b::MI355XBLAS4, d_rays::Ptr{RTRay}, d_hits::Ptr{RTHitResult}, n::Integer; any::Bool = false, stream::Ptr{Cvoid} = C_NULL) =
    check(any ? ccall((:rc_trace_any4_device, LIB), Cint, (Ptr{Cvoid}, UInt32, Ptr{RTRay}, Ptr{RTHitResult}, UInt64, Ptr{Cvoid}), b.scene.ptr, b.blas_id, d_rays, d_hits, n, stream) :
                ccall((:rc_trace_closest4_device, LIB), Cint, (Ptr{Cvoid}, UInt32, Ptr{RTRay}, Ptr{RTHitResult}, UInt64, Ptr{Cvoid}), b.scene.ptr, b.blas_id, d_rays, d_hits, n, stream))

end # module
