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
# every non-zero status becomes ErrorException, the type the reference's tests expect
# (test/test_tlas_stress.jl:585-617: @test_throws ErrorException update_transform!(tlas, deleted_handle, ...))
check(status::Cint) = status == 0 ? nothing : error(last_error())

"Mutable accel: plays the role of Raycore.TLAS{Backend} (src/instanced-bvh.jl:261-310)."
mutable struct MI355XTLAS <: AbstractAccel
    backend::MI355XBackend
    ptr::Ptr{Cvoid}
    prims::Vector{Triangle{UInt32}}      # host copy of all_blas_prims (Morton-sorted), refreshed after a rebuild
    prims_valid::Bool
    function MI355XTLAS(backend::MI355XBackend = MI355XBackend())
        ref = Ref{Ptr{Cvoid}}(C_NULL)
        check(ccall((:rc_scene_create, LIB), Cint, (Cint, Ref{Ptr{Cvoid}}), backend.device, ref))   # TLAS(backend), :334-358
        tlas = new(backend, ref[], Triangle{UInt32}[], false)
        finalizer(Raycore.free!, tlas)
        return tlas
    end
end

"The adapted form (StaticTLAS analogue, src/instanced-bvh.jl:155-168): what gets passed to trace calls."
struct MI355XStaticTLAS <: AbstractAdaptedAccel
    owner::MI355XTLAS
end

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
    return (true, primitives(a)[h.primitive_id + 1], h.t, SVector{3, Float32}(w, h.bary_u, h.bary_v), h.instance_id + UInt32(1))
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
function Raycore.view_factors(a::MI355XStaticTLAS; rays_per_triangle = 10000, seed::UInt64 = rand(UInt64))
    n = counts(a.owner)[4]
    out = Matrix{UInt32}(undef, n, n)      # column-major, [src_meta, hit_meta] as in the reference
    check(ccall((:rc_view_factors, LIB), Cint, (Ptr{Cvoid}, UInt32, UInt64, Ptr{UInt32}), a.owner.ptr, rays_per_triangle, seed, out))
    return out
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

end # module
