// rc_drivers.hip -- the analysis drivers that sit directly on closest_hit, run on the device.
//
// Replaces src/kernels.jl: generate_ray_grid (:10-56), hits_from_grid (:58-72), get_illumination (:112-124)
// and view_factors / view_factors! (:74-104) with their sampling helpers random_triangle_point,
// random_hemisphere_uniform, get_orthogonal_basis (src/math.jl:125-174).  The reference runs these as
// Threads.@threads CPU loops over closest_hit with a serial Dict histogram; here rays are generated in the
// kernel that traces them (no ray I/O) and results are accumulated with integer-valued device atomics,
// which are exact and order-independent.
//
// Determinism: the reference's view_factors uses the unseeded task-local RNG.  Here every (source
// primitive, ray index) pair draws its four uniforms from Philox4x32-10 keyed by the seed, so a result does
// not depend on the launch geometry or on how the job is sharded over GPUs.  sin/cos/acos are evaluated in
// f64 with fixed-order polynomial kernels (fdlibm coefficients) and rounded once to f32 -- Julia's Float32
// trig also evaluates in higher precision and rounds once -- so the CPU oracle, which uses the same formulas,
// generates bit-identical rays.
#include <hipcub/hipcub.hpp>

#include "rc_traverse_core.h"

#include <cmath>

namespace {

using namespace rc;

struct GridParams {  // everything generate_ray_grid computes before its double loop
    float gc[3], b1[3], b2[3], dir[3];
    float cell_w, cell_h;
    uint32_t grid;
};

// generate_ray_grid body (:47-55): u, v and the sum are Float64 ((grid_size+1)/2 is Float64) and round
// once into the Float32 point.
__device__ inline RcRay grid_ray(const GridParams& g, uint64_t idx) {
    uint32_t i = (uint32_t)(idx % g.grid) + 1, j = (uint32_t)(idx / g.grid) + 1;
    double half = ((double)g.grid + 1.0) / 2.0;
    double u = ((double)i - half) * (double)g.cell_w;
    double v = ((double)j - half) * (double)g.cell_h;
    RcRay r;
    r.ox = (float)(((double)g.gc[0] + u * (double)g.b1[0]) + v * (double)g.b2[0]);
    r.oy = (float)(((double)g.gc[1] + u * (double)g.b1[1]) + v * (double)g.b2[1]);
    r.oz = (float)(((double)g.gc[2] + u * (double)g.b1[2]) + v * (double)g.b2[2]);
    r.tmin = 0.0f; r.dx = g.dir[0]; r.dy = g.dir[1]; r.dz = g.dir[2]; r.tmax = INFINITY;
    return r;
}

__global__ __launch_bounds__(kBlock) void k_ray_grid(GridParams g, RcRay* out) {
    uint64_t n = (uint64_t)g.grid * g.grid;
    for (uint64_t i = blockIdx.x * (uint64_t)kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
        RcRay r = grid_ray(g, i);
        float4* q = reinterpret_cast<float4*>(out + i);
        q[0] = make_float4(r.ox, r.oy, r.oz, r.tmin);
        q[1] = make_float4(r.dx, r.dy, r.dz, r.tmax);
    }
}

// hits_from_grid + the histogram of get_illumination (:58-72, :112-124) fused on the persistent phased traversal
// core: the ray is generated when a lane is refilled, and a finished ray does one f32 atomic add of 1.0 on the hit
// primitive's metadata slot (exact while counts < 2^24, as in the reference's Float32 sums).
struct GridSource {
    GridParams g;
    uint64_t first;
    __device__ inline RcRay operator()(uint64_t i) const { return grid_ray(g, first + i); }
    static constexpr bool kPrefetch = false;  // rays are generated, nothing to read ahead (ArraySource::prefetch)
};
// Counting into an accumulator array from inside a wave.  An atomic on ONE address costs ~12.7 ns on this chip whether or not it
// returns a value and however many lanes issue it (tools/archive/atomic_probe.hip), so a driver whose rays mostly land on a few large
// triangles (a ground plane under get_illumination: 10^5 hits on one counter = milliseconds) would be bound by that one word.
// The lanes that finish together first combine equal targets: up to four rounds of "the first pending lane's target, everyone
// with the same target, one atomic of the group's size", stopping at the first group of one (the common case of all-different
// targets then costs a single round); whatever is still pending adds its own 1.  Called by the active (finishing, counted) lanes only.
template <class T>
__device__ inline void wave_count(T* acc, unsigned long long index, bool counted) {
    bool pending = counted;
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
        const unsigned long long act = __ballot(pending);
        if (!act) return;
        const int leader = __ffsll((long long)act) - 1;
        const unsigned lo = __shfl((unsigned)index, leader), hi = __shfl((unsigned)(index >> 32), leader);
        const bool same = pending && index == (((unsigned long long)hi << 32) | lo);
        const unsigned long long grp = __ballot(same);
        if ((int)(threadIdx.x & 63u) == leader) atomicAdd(acc + index, (T)__popcll(grp));
        pending = pending && !same;
        if (__popcll(grp) == 1) break;  // a group of one: the targets are probably all different, stop looking for duplicates
    }
    if (pending) atomicAdd(acc + index, (T)1);
}

// get_illumination's histogram is a HOT vector (a few large, well-lit triangles take most of the hits: C5 seen from outside counts 6 852 rays on
// one wall triangle), and atomics onto a few cache lines serialise: the call took 0.59 ms where tracing the same rays takes 0.19.  The kernel
// counts into kHistCopies private u32 copies (by workgroup), each laid out transposed in rows of eight, and k_illumination_fold adds them into
// the caller's f32 vector with the reference's arithmetic: `counts[k] += 1f0` per ray (src/kernels.jl:119-121) is exact up to 2^24 and stays
// there, so counts = min(counts + hits, 2^24) in integers is the same number.  (Also tried: the traversal only STORES each ray's hit
// metadata and a streaming kernel counts afterwards -- two more launches and a 16 MB round trip: C3 0.62 ms instead of 0.57, C5 0.34 instead of 0.24.)
constexpr uint32_t kHistCopies = 16;
struct HistogramSink {
    const RcInstRec* inst;
    const RcPrim* prims;
    uint32_t n_prims;
    uint32_t* scratch;  // kHistCopies x 8 x n8 counters, zeroed by the launcher
    uint32_t n8;        // ceil(n_prims / 8)
    __device__ inline void operator()(uint64_t, bool hit, float, float, float, uint32_t prim, int instance) const {
        uint32_t meta = 0;
        if (hit) {
            const uint4 m3 = *(reinterpret_cast<const uint4*>(inst + instance) + 3);
            meta = prims[m3.y + prim - 1u].meta;
        }
        const bool counted = hit && meta >= 1 && meta <= n_prims;  // metadata outside 1..N is dropped (src/kernels.jl:123)
        const uint32_t j = meta - 1u;
        wave_count(scratch + (size_t)(blockIdx.x & (kHistCopies - 1u)) * 8u * n8, counted ? (unsigned long long)((j & 7u) * n8 + (j >> 3)) : 0ull, counted);
    }
};
__global__ void k_illumination_fold(const uint32_t* scratch, uint32_t n8, uint32_t n, float* counts) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    unsigned long long sum = 0;
    for (uint32_t c = 0; c < kHistCopies; ++c) sum += scratch[(size_t)c * 8u * n8 + (j & 7u) * n8 + (j >> 3)];
    if (sum == 0) return;
    // ACCUMULATES, atomically (shards traced concurrently on several streams into ONE vector used to be safe when the kernel itself
    // added with atomics; N compare-and-swaps per launch cost nothing): counts = min(counts + hits, 2^24) in integers
    uint32_t* word = reinterpret_cast<uint32_t*>(counts + j);
    uint32_t seen = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (;;) {
        const float before = __uint_as_float(seen);  // an integer below 2^24 unless the caller put something else there
        float after;
        if (before >= 0.0f && before <= 16777216.0f && before == (float)(uint32_t)before) {
            const unsigned long long total = (unsigned long long)(uint32_t)before + sum;
            after = total >= 16777216ull ? 16777216.0f : (float)(uint32_t)total;
        } else {  // not a count: add one by one like the reference would, up to where f32 stops moving
            after = before;
            for (unsigned long long k = 0; k < sum && k < 16777216ull; ++k) { const float nv = after + 1.0f; if (nv == after) break; after = nv; }
        }
        const uint32_t prev = atomicCAS(word, seen, __float_as_uint(after));
        if (prev == seen) break;
        seen = prev;
    }
}
__global__ __launch_bounds__(kBlock, 6) void k_illumination(SceneView v, PersistArgs p, GridParams g, uint64_t ray_begin, uint32_t* scratch, uint32_t n8) {
    __shared__ uint32_t lds_stack[kLdsStack * kBlock];
    phased_trace<false, kLdsStack, false>(v, p, lds_stack, GridSource{g, ray_begin}, HistogramSink{v.inst, v.prims, v.n_prims, scratch, n8});
}
// Small top level (<= kTlasLdsNodes nodes): the shape of trace kernel 5 -- two 768-thread workgroups per CU, TLAS (and a single BLAS's
// top nodes) in LDS planes.
__global__ __launch_bounds__(kMidBlock, 6) void k_illumination_lds(SceneView v, PersistArgs p, GridParams g, uint64_t ray_begin, uint32_t* scratch, uint32_t n8) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const LdsTop top(smem + (size_t)kMidStack * kMidBlock * 4);
    stage_lds_top<kMidBlock>(top, v, p.blas_k, p.lds_blas_base);
    __syncthreads();
    phased_trace<false, kMidStack, false, GridSource, HistogramSink, kMidBlock, true, true>(v, p, reinterpret_cast<uint32_t*>(smem), GridSource{g, ray_begin},
                                                                                            HistogramSink{v.inst, v.prims, v.n_prims, scratch, n8}, top);
}
// Larger top levels (the shape of trace kernel 6): only the breadth-first tops of the TLAS and of a single BLAS are staged.
__global__ __launch_bounds__(kMidBlock, 6) void k_illumination_partial(SceneView v, PersistArgs p, GridParams g, uint64_t ray_begin, uint32_t* scratch, uint32_t n8) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LdsTop top;
    top.tl = reinterpret_cast<float2*>(smem + (size_t)kMidStack * kMidBlock * 4);
    stage_partial_top<kMidBlock>(top.tl, v, p.tlas_k, p.blas_k, p.lds_blas_base);
    __syncthreads();
    phased_trace<false, kMidStack, false, GridSource, HistogramSink, kMidBlock, false, false, true>(v, p, reinterpret_cast<uint32_t*>(smem), GridSource{g, ray_begin},
                                                                                                     HistogramSink{v.inst, v.prims, v.n_prims, scratch, n8}, top);
}

// ---- Philox4x32-10 (Salmon et al., SC'11) ---------------------------------------------------------------
__device__ inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ inline float u32_to_unit(uint32_t x) { return (float)(x >> 8) * 0x1.0p-24f; }

// ---- f64 trig with a fixed operation order (fdlibm kernels), valid for the ranges the sampler needs -----
__device__ inline void sincos_f64(double x, double& s, double& c) {  // 0 <= x <= 2 pi
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17, two_over_pi = 6.36619772367581382433e-01;
    int k = (int)(x * two_over_pi + 0.5);
    double r = (x - (double)k * pio2_hi) - (double)k * pio2_lo;
    double z = r * r;
    double sp = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    double ks = r + (z * r) * (-1.66666666666666324348e-01 + z * sp);
    double cp = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    double kc = 1.0 - (0.5 * z - z * cp);
    switch (k & 3) {
        case 0: s = ks; c = kc; break;
        case 1: s = kc; c = -ks; break;
        case 2: s = -ks; c = -kc; break;
        default: s = -kc; c = ks; break;
    }
}
__device__ inline double acos_f64(double x) {  // 0 <= x < 1
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
    const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01, pS2 = 2.01212532134862925881e-01,
                 pS3 = -4.00555345006794114027e-02, pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
                 qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00, qS3 = -6.88283971605453293030e-01,
                 qS4 = 7.70381505559019352791e-02;
    if (x < 0.5) {
        double z = x * x;
        double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        double r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    double z = (1.0 - x) * 0.5;
    double s = __builtin_sqrt(z);  // IEEE f64 square root
    double df = __longlong_as_double(__double_as_longlong(s) & 0xFFFFFFFF00000000ll);
    double c = (z - df * df) / (s + df);
    double p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    double q = 1.0 + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    double r = p / q;
    double w = r * s + c;
    return 2.0 * (df + w);
}

// norm(a) = sqrt(dot(a,a)); normalize(a) = a ./ norm(a) (GeometryBasics 0.5 fixed_arrays).
// sqrtf and / are the IEEE correctly-rounded forms here (-fhip-fp32-correctly-rounded-divide-sqrt); HIP's __fsqrt_rn is
// the NATIVE (approximate) square root unless OCML_BASIC_ROUNDED_OPERATIONS is defined, so it is not used.
__device__ inline float3_ normalize3(float3_ a) {
    float n = __builtin_sqrtf(dot3(a, a));
    return mk3((a.x / n), (a.y / n), (a.z / n));
}

// The ray view_factors! shoots for (source primitive, ray index) (:83-92 + src/math.jl:125-174)
__device__ inline RcRay view_factor_ray(const RcPrim& tri, uint32_t src, uint32_t ray_idx, uint32_t k0, uint32_t k1) {
    float3_ p1 = mk3(tri.v[0], tri.v[1], tri.v[2]), p2 = mk3(tri.v[3], tri.v[4], tri.v[5]), p3 = mk3(tri.v[6], tri.v[7], tri.v[8]);
    float3_ normal = normalize3(cross3(sub3(p2, p1), sub3(p3, p1)));  // GB.orthogonal_vector + normalize (:86-87)
    // get_orthogonal_basis (src/math.jl:143-156)
    float3_ n = normalize3(normal);
    float ax = fabsf(normal.x), ay = fabsf(normal.y), az = fabsf(normal.z);
    int mi = 1; float mv = ax;
    if (ay < mv) { mi = 2; mv = ay; }
    if (az < mv) { mi = 3; mv = az; }
    float3_ cand = mi == 1 ? mk3(1, 0, 0) : (mi == 2 ? mk3(0, 1, 0) : mk3(0, 0, 1));
    float3_ bv = normalize3(cross3(n, cand));
    float3_ bu = normalize3(cross3(bv, n));
    uint32_t rnd[4];
    philox4x32_10(ray_idx, src, 0u, 0u, k0, k1, rnd);
    float r1 = u32_to_unit(rnd[0]), r2 = u32_to_unit(rnd[1]), xi1 = u32_to_unit(rnd[2]), xi2 = u32_to_unit(rnd[3]);
    // random_triangle_point (src/math.jl:158-174)
    float sqrt_r1 = __builtin_sqrtf(r1);
    float wu = 1.0f - sqrt_r1, wv = sqrt_r1 * (1.0f - r2), ww = sqrt_r1 * r2;
    float3_ pt = add3(add3(scale3(p1, wu), scale3(p2, wv)), scale3(p3, ww));
    float3_ o = add3(pt, scale3(normal, 0.01f));  // :91
    // random_hemisphere_uniform (src/math.jl:125-141)
    float theta = (float)acos_f64((double)xi1);
    float phi = (2.0f * 3.1415927f) * xi2;
    double st, ct, sp, cp;
    sincos_f64((double)theta, st, ct);
    sincos_f64((double)phi, sp, cp);
    float sin_t = (float)st, cos_t = (float)ct, sin_p = (float)sp, cos_p = (float)cp;
    float xl = sin_t * cos_p, yl = sin_t * sin_p, zl = cos_t;
    float3_ d = add3(add3(scale3(bu, xl), scale3(bv, yl)), scale3(normal, zl));
    return RcRay{o.x, o.y, o.z, 0.0f, d.x, d.y, d.z, INFINITY};
}

// view_factors! (:80-104): work item = (source primitive, ray); result[src_meta, hit_meta] += 1 when the hit
// primitive's metadata differs.  One u32 atomic per counted ray.  Runs on the persistent phased traversal core.
struct ViewFactorSource {
    const RcPrim* prims;
    const uint32_t* order;  // RC_VF_SOURCES_BY_METADATA: source position -> flat primitive index, else nullptr (position = index)
    uint32_t k0, k1, src_begin, ray_begin, n_ray;
    __device__ inline RcRay operator()(uint64_t w) const {
        const uint32_t pos = src_begin + (uint32_t)(w / n_ray), ray_idx = ray_begin + (uint32_t)(w % n_ray);
        const uint32_t src = order ? order[pos] : pos;
        return view_factor_ray(prims[src], src, ray_idx, k0, k1);  // Philox is keyed by the PRIMITIVE index: the rays do not depend on the addressing
    }
    static constexpr bool kPrefetch = false;
};
struct ViewFactorSink {
    const RcInstRec* inst;
    const RcPrim* prims;
    const uint32_t* order;
    uint32_t n_prims, src_begin, n_ray;
    uint32_t* matrix;
    uint64_t row_stride, col_stride;
    uint32_t row_offset, flags;
    __device__ inline void operator()(uint64_t w, bool hit, float, float, float, uint32_t prim, int instance) const {
        bool counted = false;
        unsigned long long index = 0;
        if (hit) {
            const uint32_t pos = src_begin + (uint32_t)(w / n_ray);
            const uint32_t src = order ? order[pos] : pos;
            const uint4 m3 = *(reinterpret_cast<const uint4*>(inst + instance) + 3);
            const uint32_t hit_meta = prims[m3.y + prim - 1u].meta, src_meta = prims[src].meta;
            counted = hit_meta != src_meta && src_meta >= 1 && src_meta <= n_prims && hit_meta >= 1 && hit_meta <= n_prims;
            const uint32_t row = (flags & 4u) ? src_meta - 1 : ((flags & 2u) ? pos : ((flags & 1u) ? src : src_meta - 1));
            if (counted) index = (uint64_t)(row - row_offset) * row_stride + (uint64_t)(hit_meta - 1) * col_stride;
        }
        wave_count(matrix, index, counted);
    }
};
// Per-triangle TOTALS of the same job (rc_view_factor_totals*): the column sums received[j] = sum_i result[i, j] -- what the reference's own
// use of the matrix reads off it (docs/src/viewfactors_content.md:62-68: sum(view(viewf_matrix, :, i))) -- and the row sums emitted[i] = sum_j
// result[i, j], accumulated directly: two u64 atomics per counted ray, no N x N array anywhere.  Same rays, same counting rule as
// ViewFactorSink (:93-97); 64-bit so that N x rays_per_triangle may pass 2^32.
// `received` is the hot vector: most rays of a closed scene end on a few large triangles, and 2 x 10^8 atomics onto a few hundred cache
// lines ran at HALF the rate of the matrix's scattered ones (C5: 81 ms against 39).  The kernel therefore counts into kTotalsCopies private
// copies (by workgroup), each laid out TRANSPOSED in rows of eight (neighbouring triangles -- a wall's two halves, a sphere's ring -- sit in
// different cache lines), and k_vf_totals_fold adds the copies into the caller's vector afterwards.
constexpr uint32_t kTotalsCopies = 16;
struct ViewFactorTotalsSink {
    const RcInstRec* inst;
    const RcPrim* prims;
    uint32_t n_prims, src_begin, n_ray;
    unsigned long long* received;  // kTotalsCopies x 8 x n8 scratch counters (zeroed by the launcher), or nullptr
    uint32_t n8;                   // ceil(n_prims / 8)
    unsigned long long* emitted;   // [n_prims] or nullptr: per source, nearly wave-uniform -- one atomic per group (wave_count)
    __device__ inline void operator()(uint64_t w, bool hit, float, float, float, uint32_t prim, int instance) const {
        bool counted = false;
        uint32_t hit_meta = 1, src_meta = 1;
        if (hit) {
            const uint32_t src = src_begin + (uint32_t)(w / n_ray);
            const uint4 m3 = *(reinterpret_cast<const uint4*>(inst + instance) + 3);
            hit_meta = prims[m3.y + prim - 1u].meta; src_meta = prims[src].meta;
            counted = hit_meta != src_meta && src_meta >= 1 && src_meta <= n_prims && hit_meta >= 1 && hit_meta <= n_prims;
        }
        if (received) {
            const uint32_t j = hit_meta - 1u;
            wave_count(received + (size_t)(blockIdx.x & (kTotalsCopies - 1u)) * 8u * n8, (unsigned long long)((j & 7u) * n8 + (j >> 3)), counted);
        }
        if (emitted) wave_count(emitted, (unsigned long long)(src_meta - 1u), counted);
    }
};
__global__ void k_vf_totals_fold(const unsigned long long* scratch, uint32_t n8, uint32_t n, unsigned long long* received) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    unsigned long long sum = 0;
    for (uint32_t c = 0; c < kTotalsCopies; ++c) sum += scratch[(size_t)c * 8u * n8 + (j & 7u) * n8 + (j >> 3)];
    if (sum) atomicAdd(received + j, sum);  // (ACCUMULATES; atomic, so that shards traced concurrently on several streams may share one vector -- ADVICE r4)
}
__global__ __launch_bounds__(kBlock, 6) void k_vf_totals(SceneView v, PersistArgs p, uint32_t k0, uint32_t k1, uint32_t src_begin, uint32_t ray_begin,
                                                          uint32_t n_ray, unsigned long long* received, uint32_t n8, unsigned long long* emitted) {
    __shared__ uint32_t lds_stack[kLdsStack * kBlock];
    phased_trace<false, kLdsStack, false>(v, p, lds_stack, ViewFactorSource{v.prims, nullptr, k0, k1, src_begin, ray_begin, n_ray},
                                          ViewFactorTotalsSink{v.inst, v.prims, v.n_prims, src_begin, n_ray, received, n8, emitted});
}
__global__ __launch_bounds__(kMidBlock, 6) void k_vf_totals_lds(SceneView v, PersistArgs p, uint32_t k0, uint32_t k1, uint32_t src_begin, uint32_t ray_begin,
                                                                 uint32_t n_ray, unsigned long long* received, uint32_t n8, unsigned long long* emitted) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const LdsTop top(smem + (size_t)kMidStack * kMidBlock * 4);
    stage_lds_top<kMidBlock>(top, v, p.blas_k, p.lds_blas_base);
    __syncthreads();
    phased_trace<false, kMidStack, false, ViewFactorSource, ViewFactorTotalsSink, kMidBlock, true, true>(
        v, p, reinterpret_cast<uint32_t*>(smem), ViewFactorSource{v.prims, nullptr, k0, k1, src_begin, ray_begin, n_ray},
        ViewFactorTotalsSink{v.inst, v.prims, v.n_prims, src_begin, n_ray, received, n8, emitted}, top);
}
__global__ __launch_bounds__(kMidBlock, 6) void k_vf_totals_partial(SceneView v, PersistArgs p, uint32_t k0, uint32_t k1, uint32_t src_begin, uint32_t ray_begin,
                                                                     uint32_t n_ray, unsigned long long* received, uint32_t n8, unsigned long long* emitted) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LdsTop top;
    top.tl = reinterpret_cast<float2*>(smem + (size_t)kMidStack * kMidBlock * 4);
    stage_partial_top<kMidBlock>(top.tl, v, p.tlas_k, p.blas_k, p.lds_blas_base);
    __syncthreads();
    phased_trace<false, kMidStack, false, ViewFactorSource, ViewFactorTotalsSink, kMidBlock, false, false, true>(
        v, p, reinterpret_cast<uint32_t*>(smem), ViewFactorSource{v.prims, nullptr, k0, k1, src_begin, ray_begin, n_ray},
        ViewFactorTotalsSink{v.inst, v.prims, v.n_prims, src_begin, n_ray, received, n8, emitted}, top);
}
__global__ __launch_bounds__(kBlock, 6) void k_view_factors(SceneView v, PersistArgs p, uint32_t k0, uint32_t k1, uint32_t src_begin,
                                                             uint32_t ray_begin, uint32_t n_ray, uint32_t* matrix, uint64_t row_stride,
                                                             uint64_t col_stride, uint32_t row_offset, uint32_t flags, const uint32_t* order) {
    __shared__ uint32_t lds_stack[kLdsStack * kBlock];
    phased_trace<false, kLdsStack, false>(v, p, lds_stack, ViewFactorSource{v.prims, order, k0, k1, src_begin, ray_begin, n_ray},
                                          ViewFactorSink{v.inst, v.prims, order, v.n_prims, src_begin, n_ray, matrix, row_stride, col_stride, row_offset, flags});
}
__global__ __launch_bounds__(kMidBlock, 6) void k_view_factors_lds(SceneView v, PersistArgs p, uint32_t k0, uint32_t k1, uint32_t src_begin,
                                                                    uint32_t ray_begin, uint32_t n_ray, uint32_t* matrix, uint64_t row_stride,
                                                                    uint64_t col_stride, uint32_t row_offset, uint32_t flags, const uint32_t* order) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const LdsTop top(smem + (size_t)kMidStack * kMidBlock * 4);
    stage_lds_top<kMidBlock>(top, v, p.blas_k, p.lds_blas_base);
    __syncthreads();
    phased_trace<false, kMidStack, false, ViewFactorSource, ViewFactorSink, kMidBlock, true, true>(
        v, p, reinterpret_cast<uint32_t*>(smem), ViewFactorSource{v.prims, order, k0, k1, src_begin, ray_begin, n_ray},
        ViewFactorSink{v.inst, v.prims, order, v.n_prims, src_begin, n_ray, matrix, row_stride, col_stride, row_offset, flags}, top);
}
__global__ __launch_bounds__(kMidBlock, 6) void k_view_factors_partial(SceneView v, PersistArgs p, uint32_t k0, uint32_t k1, uint32_t src_begin,
                                                                        uint32_t ray_begin, uint32_t n_ray, uint32_t* matrix, uint64_t row_stride,
                                                                        uint64_t col_stride, uint32_t row_offset, uint32_t flags, const uint32_t* order) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LdsTop top;
    top.tl = reinterpret_cast<float2*>(smem + (size_t)kMidStack * kMidBlock * 4);
    stage_partial_top<kMidBlock>(top.tl, v, p.tlas_k, p.blas_k, p.lds_blas_base);
    __syncthreads();
    phased_trace<false, kMidStack, false, ViewFactorSource, ViewFactorSink, kMidBlock, false, false, true>(
        v, p, reinterpret_cast<uint32_t*>(smem), ViewFactorSource{v.prims, order, k0, k1, src_begin, ray_begin, n_ray},
        ViewFactorSink{v.inst, v.prims, order, v.n_prims, src_begin, n_ray, matrix, row_stride, col_stride, row_offset, flags}, top);
}

__global__ void k_meta_keys(const RcPrim* prims, uint32_t n, uint32_t* keys, uint32_t* vals) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { keys[i] = prims[i].meta; vals[i] = i; }
}

__global__ void k_view_factor_rays(SceneView v, uint32_t k0, uint32_t k1, uint32_t src, uint32_t ray_begin, uint32_t n_ray, RcRay* out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_ray) out[i] = view_factor_ray(v.prims[src], src, ray_begin + i, k0, k1);
}

// ---- wavefront stages adjacent to the trace (docs/src/wavefront-renderer.jl:296-333) ---------
// Geometric world-space normal of a hit: normalize(cross(v1-v0, v2-v0)) of the hit primitive carried through the
// instance's inverse-transpose, flipped to face the ray origin.
__device__ inline void hit_frame(const SceneView& v, const RcRay& r, const RcHit& h, float3_& point, float3_& normal) {
    point = add3(mk3(r.ox, r.oy, r.oz), scale3(mk3(r.dx, r.dy, r.dz), h.t));  // hit_point = ray.o + ray.d * dist (:302)
    const RcPrim tri = v.prims[h.primitive_id];
    const float3_ v0 = mk3(tri.v[0], tri.v[1], tri.v[2]), v1 = mk3(tri.v[3], tri.v[4], tri.v[5]), v2 = mk3(tri.v[6], tri.v[7], tri.v[8]);
    const float3_ nl = cross3(sub3(v1, v0), sub3(v2, v0));
    const float* m = v.inst[h.instance_id].inv;  // n_w = transpose(inv 3x3) * n_l
    float3_ nw = mk3(m[0] * nl.x + m[4] * nl.y + m[8] * nl.z, m[1] * nl.x + m[5] * nl.y + m[9] * nl.z, m[2] * nl.x + m[6] * nl.y + m[10] * nl.z);
    nw = normalize3(nw);
    if (dot3(nw, mk3(r.dx, r.dy, r.dz)) > 0.0f) nw = mk3(-nw.x, -nw.y, -nw.z);
    normal = nw;
}

__global__ void k_hit_points(SceneView v, const RcRay* rays, const RcHit* hits, uint64_t n, float* points, float* normals) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        float3_ p = mk3(0, 0, 0), nn = mk3(0, 0, 0);
        const RcHit h = hits[i];
        if (h.hit) hit_frame(v, rays[i], h, p, nn);
        points[3 * i] = p.x; points[3 * i + 1] = p.y; points[3 * i + 2] = p.z;
        if (normals) { normals[3 * i] = nn.x; normals[3 * i + 1] = nn.y; normals[3 * i + 2] = nn.z; }
    }
}

// generate_shadow_rays! for one point light (:288-333): slot i of the output belongs to ray i; misses get the
// reference's dummy ray (o = 0, d = (0,0,1), t_max = 0), hits a ray from hit_point + normal*bias toward the light with
// t_max = distance, ready for rc_trace_any_device.
__global__ void k_shadow_rays(SceneView v, const RcRay* rays, const RcHit* hits, uint64_t n, float lx, float ly, float lz, float bias, RcRay* out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        RcRay s{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f};
        const RcHit h = hits[i];
        if (h.hit) {
            float3_ p, nn;
            hit_frame(v, rays[i], h, p, nn);
            const float3_ o = add3(p, scale3(nn, bias));
            const float3_ lv = sub3(mk3(lx, ly, lz), o);
            const float dist = __builtin_sqrtf(dot3(lv, lv));
            s = RcRay{o.x, o.y, o.z, 0.f, (lv.x / dist), (lv.y / dist), (lv.z / dist), dist};
        }
        float4* q = reinterpret_cast<float4*>(out + i);
        q[0] = make_float4(s.ox, s.oy, s.oz, s.tmin);
        q[1] = make_float4(s.dx, s.dy, s.dz, s.tmax);
    }
}

// generate_primary_rays_lookat! (docs/src/wavefront-renderer.jl:219-254): ray (pixel_idx-1)*samples + s for pixel (x, y)
// (1-based, row-major pixel_idx = (y-1)*width + x); jitter = rand(Vec2f) there, Philox4x32-10(seed; ray index) here, or the
// pixel centre when jitter is off.  u, v as written: 2*(x - 0.5 + j1)/width - 1 and 1 - 2*(y - 0.5 + j2)/height.
struct CameraParams {
    float pos[3], right[3], up[3], forward[3];
    float half_width, half_height;
    uint32_t width, height, samples, jitter;
    uint32_t k0, k1;
};
__global__ void k_primary_rays(CameraParams c, uint64_t n, RcRay* out) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t pixel = i / c.samples;
        const uint32_t x = (uint32_t)(pixel % c.width) + 1u, y = (uint32_t)(pixel / c.width) + 1u;
        float j1 = 0.5f, j2 = 0.5f;
        if (c.jitter) {
            uint32_t rnd[4];
            philox4x32_10((uint32_t)i, (uint32_t)(i >> 32), 0u, 0x50524159u, c.k0, c.k1, rnd);
            j1 = u32_to_unit(rnd[0]); j2 = u32_to_unit(rnd[1]);
        }
        const float u = 2.0f * ((float)x - 0.5f + j1) / (float)c.width - 1.0f;
        const float v = 1.0f - 2.0f * ((float)y - 0.5f + j2) / (float)c.height;
        const float su = u * c.half_width, sv = v * c.half_height;
        const float3_ d = normalize3(add3(add3(mk3(c.forward[0], c.forward[1], c.forward[2]), scale3(mk3(c.right[0], c.right[1], c.right[2]), su)),
                                          scale3(mk3(c.up[0], c.up[1], c.up[2]), sv)));
        float4* q = reinterpret_cast<float4*>(out + i);
        q[0] = make_float4(c.pos[0], c.pos[1], c.pos[2], 0.0f);
        q[1] = make_float4(d.x, d.y, d.z, INFINITY);  // Ray(o=..., d=...): t_min 0, t_max Inf (src/ray.jl:1-7)
    }
}

__global__ void k_hit_flags(const RcHit* hits, uint64_t n, uint32_t* flags) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i < n) flags[i] = hits[i].hit ? 1u : 0u;
}
__global__ void k_scatter_hit_indices(const uint32_t* flags, const uint32_t* pos, uint64_t n, uint32_t* indices, uint32_t* count) {
    const uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flags[i]) indices[pos[i]] = (uint32_t)i;
    if (i == n - 1) *count = pos[i] + flags[i];
}

float3_ h_normalize(float3_ a) {
    float n = sqrtf(dot3(a, a));
    return mk3(a.x / n, a.y / n, a.z / n);
}

// generate_ray_grid up to its loop (:10-46), as called from hits_from_grid (:58-61).  Host code compiled
// with -ffp-contract=off; jl_min/jl_max give extrema() its Julia semantics.
GridParams grid_params(rc_scene* s, const float viewdir[3], uint32_t grid) {
    float3_ ray_direction = h_normalize(mk3(viewdir[0], viewdir[1], viewdir[2]));
    float3_ direction = h_normalize(ray_direction);
    float3_ o = mk3(s->root_min[0], s->root_min[1], s->root_min[2]);
    float3_ w = sub3(mk3(s->root_max[0], s->root_max[1], s->root_max[2]), o);
    float3_ temp = fabsf(direction.x) < 0.9f ? mk3(1, 0, 0) : mk3(0, 1, 0);
    float3_ b1 = h_normalize(cross3(direction, temp));
    float3_ b2 = h_normalize(cross3(direction, b1));
    float min1 = INFINITY, max1 = -INFINITY, min2 = INFINITY, max2 = -INFINITY, mind = INFINITY;
    for (int c = 0; c < 8; ++c) {
        float3_ p = mk3(o.x + ((c & 1) ? 1.0f : 0.0f) * w.x, o.y + ((c & 2) ? 1.0f : 0.0f) * w.y, o.z + ((c & 4) ? 1.0f : 0.0f) * w.z);
        float p1 = dot3(p, b1), p2 = dot3(p, b2), pd = dot3(p, direction);
        min1 = jl_min(min1, p1); max1 = jl_max(max1, p1);
        min2 = jl_min(min2, p2); max2 = jl_max(max2, p2);
        mind = jl_min(mind, pd);
    }
    float margin = 0.05f * jl_max(max1 - min1, max2 - min2);
    float grid_width = max1 - min1 + 2.0f * margin, grid_height = max2 - min2 + 2.0f * margin;
    float min_depth = mind - margin;
    float c1 = (min1 + max1) / 2.0f, c2 = (min2 + max2) / 2.0f;
    float3_ gc = add3(add3(add3(mk3(0, 0, 0), scale3(direction, min_depth)), scale3(b1, c1)), scale3(b2, c2));
    GridParams g;
    g.gc[0] = gc.x; g.gc[1] = gc.y; g.gc[2] = gc.z;
    g.b1[0] = b1.x; g.b1[1] = b1.y; g.b1[2] = b1.z;
    g.b2[0] = b2.x; g.b2[1] = b2.y; g.b2[2] = b2.z;
    g.dir[0] = ray_direction.x; g.dir[1] = ray_direction.y; g.dir[2] = ray_direction.z;
    g.cell_w = grid_width / (float)grid; g.cell_h = grid_height / (float)grid;
    g.grid = grid;
    return g;
}

}  // namespace

void rc_launch_ray_grid(rc_scene* s, const float viewdir[3], uint32_t grid, RcRay* d_rays, hipStream_t stream) {
    GridParams g = grid_params(s, viewdir, grid);
    uint64_t n = (uint64_t)grid * grid;
    uint32_t blocks = (uint32_t)std::min<uint64_t>((n + kBlock - 1) / kBlock, (uint64_t)s->n_cus * 8);
    if (blocks == 0) return;
    hipLaunchKernelGGL(k_ray_grid, dim3(blocks), dim3(kBlock), 0, stream, g, d_rays);
    RC_HIP(hipGetLastError());
}

static void check_buffer_range(rc_scene* s) {
    if ((uint64_t)(s->n_flat_nodes + s->n_tlas_nodes) * 64u >= (1ull << 32))
        throw RcError(1, "driver kernels address nodes with 32-bit buffer offsets: scenes above 64 M nodes are not supported yet");
}

static unsigned long long* rc_totals_scratch(rc_scene* s, hipStream_t stream, bool capturing, size_t words);  // (below) per-stream scratch counters of the drivers

void rc_launch_illumination(rc_scene* s, const float viewdir[3], uint32_t grid, uint64_t ray_begin, uint64_t ray_end,
                            float* d_counts, hipStream_t stream) {
    if (ray_end <= ray_begin) return;
    check_buffer_range(s);
    GridParams g = grid_params(s, viewdir, grid);
    const bool partial = rc_partial_driver_ok(s);
    const bool lds = partial || rc_lds_driver_ok(s);
    const uint32_t bs = lds ? (uint32_t)kMidBlock : (uint32_t)kBlock;
    uint32_t blocks = lds ? rc_lds_driver_blocks(s, ray_end - ray_begin) : rc_persistent_blocks(s, ray_end - ray_begin);
    RcLaunchGuard launch(s, stream);
    SceneView v = rc_scene_view(s, blocks * bs);
    PersistArgs p = rc_persist_args(s, ray_end - ray_begin, blocks * bs);
    const uint32_t np = s->n_flat_prims, n8 = (np + 7u) / 8u;
    uint32_t* scratch = reinterpret_cast<uint32_t*>(rc_totals_scratch(s, stream, launch.capturing, ((size_t)kHistCopies * 8u * n8 + 1u) / 2u));  // private copies of the histogram (HistogramSink)
    RC_HIP(hipMemsetAsync(scratch, 0, sizeof(uint32_t) * kHistCopies * 8u * n8, stream));
    launch.start();
    // a repeated get_illumination (same grid: item i is the same cell every time) claims the chunks that held long rays last time first
    if (!launch.capturing && ray_begin == 0) {
        // (the rays are generated: the batch is described by its view direction and grid size -- another direction is another batch)
        const float diag = sqrtf((s->root_max[0] - s->root_min[0]) * (s->root_max[0] - s->root_min[0]) + (s->root_max[1] - s->root_min[1]) * (s->root_max[1] - s->root_min[1]) +
                                 (s->root_max[2] - s->root_min[2]) * (s->root_max[2] - s->root_min[2]));
        const float stand_in[8] = {(float)grid * diag, 0.f, 0.f, 0.f, g.dir[0], g.dir[1], g.dir[2], 0.f};
        rc_cost_order_setup(s, ray_end, 2, stream, p.claim, nullptr, stand_in);
    }
    if (partial) {
        rc_partial_driver_args(s, p);
        if (!s->lds_attr_set[8]) {
            RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_illumination_partial), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPartialLdsBytes));
            s->lds_attr_set[8] = true;
        }
        hipLaunchKernelGGL(k_illumination_partial, dim3(blocks), dim3(kMidBlock), kPartialLdsBytes, stream, v, p, g, ray_begin, scratch, n8);
    } else if (lds) {
        rc_lds_driver_args(s, p);
        if (!s->lds_attr_set[4]) {
            RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_illumination_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMidLdsBytes));
            s->lds_attr_set[4] = true;
        }
        hipLaunchKernelGGL(k_illumination_lds, dim3(blocks), dim3(kMidBlock), kMidLdsBytes, stream, v, p, g, ray_begin, scratch, n8);
    } else
    hipLaunchKernelGGL(k_illumination, dim3(blocks), dim3(kBlock), 0, stream, v, p, g, ray_begin, scratch, n8);
    if (np) hipLaunchKernelGGL(k_illumination_fold, dim3((np + 255) / 256), dim3(256), 0, stream, scratch, n8, np, d_counts);
    launch.finish();
}

// RC_VF_SOURCES_BY_METADATA: the flat primitives' indices sorted by (metadata, index) -- and the sorted metadata themselves on the host --
// cached until the next rebuild.  Built on the scene's own stream, which also owns the sort scratch.
void rc_ensure_vf_order(rc_scene* s) {
    std::lock_guard<std::mutex> g(s->launch_mu);
    if (s->vf_order_valid) return;
    const uint32_t np = s->n_flat_prims;
    s->vf_meta_sorted.assign(np, 0u);
    if (np) {
        s->keys_a.reserve(np); s->keys_b.reserve(np); s->vals_a.reserve(np); s->vf_order.reserve(np);
        hipLaunchKernelGGL(k_meta_keys, dim3((np + 255) / 256), dim3(256), 0, s->stream, s->flat_prims.p, np, s->keys_a.p, s->vals_a.p);
        size_t tmp = 0;
        RC_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, s->keys_a.p, s->keys_b.p, s->vals_a.p, s->vf_order.p, (int)np, 0, 32, s->stream));
        s->sort_tmp.reserve(tmp ? tmp : 1);
        RC_HIP(hipcub::DeviceRadixSort::SortPairs(s->sort_tmp.p, tmp, s->keys_a.p, s->keys_b.p, s->vals_a.p, s->vf_order.p, (int)np, 0, 32, s->stream));  // stable: ties keep the flat order
        RC_HIP(hipMemcpyAsync(s->vf_meta_sorted.data(), s->keys_b.p, sizeof(uint32_t) * np, hipMemcpyDeviceToHost, s->stream));
        RC_HIP(hipStreamSynchronize(s->stream));  // the launches that read the order run on other streams
    }
    s->vf_order_valid = true;
}
void rc_vf_source_range(rc_scene* s, uint32_t row_begin, uint32_t row_end, uint32_t& pos_begin, uint32_t& pos_end) {
    const auto& m = s->vf_meta_sorted;  // row r <=> metadata r + 1
    pos_begin = (uint32_t)(std::lower_bound(m.begin(), m.end(), row_begin + 1u) - m.begin());
    pos_end = (uint32_t)(std::lower_bound(m.begin(), m.end(), row_end + 1u) - m.begin());
}

void rc_launch_view_factors(rc_scene* s, uint32_t rays_per_triangle, uint64_t seed, uint32_t src_begin, uint32_t src_end,
                            uint32_t ray_begin, uint32_t ray_end, uint32_t* d_matrix, uint64_t row_stride, uint64_t col_stride,
                            uint32_t row_offset, uint32_t flags, hipStream_t stream) {
    if (src_end > s->n_flat_prims) src_end = s->n_flat_prims;
    if (ray_end > rays_per_triangle) ray_end = rays_per_triangle;
    if (src_begin >= src_end || ray_begin >= ray_end) return;
    check_buffer_range(s);
    uint64_t total = (uint64_t)(src_end - src_begin) * (ray_end - ray_begin);
    if (flags & 2u) rc_ensure_vf_order(s);  // RC_VF_SOURCES_BY_METADATA
    const uint32_t* order = (flags & 2u) ? s->vf_order.p : nullptr;
    RcLaunchGuard launch(s, stream);
    const bool partial = rc_partial_driver_ok(s);
    const bool lds = partial || rc_lds_driver_ok(s);
    const uint32_t bs = lds ? (uint32_t)kMidBlock : (uint32_t)kBlock;
    uint32_t blocks = lds ? rc_lds_driver_blocks(s, total) : rc_persistent_blocks(s, total);
    SceneView v = rc_scene_view(s, blocks * bs);
    PersistArgs p = rc_persist_args(s, total, blocks * bs);
    launch.start();
    if (partial) {
        rc_partial_driver_args(s, p);
        if (!s->lds_attr_set[9]) {
            RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_view_factors_partial), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPartialLdsBytes));
            s->lds_attr_set[9] = true;
        }
        hipLaunchKernelGGL(k_view_factors_partial, dim3(blocks), dim3(kMidBlock), kPartialLdsBytes, stream, v, p, (uint32_t)seed, (uint32_t)(seed >> 32), src_begin,
                           ray_begin, ray_end - ray_begin, d_matrix, row_stride, col_stride, row_offset, flags, order);
    } else if (lds) {
        rc_lds_driver_args(s, p);
        if (!s->lds_attr_set[5]) {
            RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_view_factors_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMidLdsBytes));
            s->lds_attr_set[5] = true;
        }
        hipLaunchKernelGGL(k_view_factors_lds, dim3(blocks), dim3(kMidBlock), kMidLdsBytes, stream, v, p, (uint32_t)seed, (uint32_t)(seed >> 32), src_begin,
                           ray_begin, ray_end - ray_begin, d_matrix, row_stride, col_stride, row_offset, flags, order);
    } else
    hipLaunchKernelGGL(k_view_factors, dim3(blocks), dim3(kBlock), 0, stream, v, p, (uint32_t)seed, (uint32_t)(seed >> 32), src_begin,
                       ray_begin, ray_end - ray_begin, d_matrix, row_stride, col_stride, row_offset, flags, order);
    launch.finish();
}

// Scratch counters of a totals launch (launch_mu held): launches on one stream are ordered and share an area; another stream gets its own (up to
// 16: then an idle stream's area is taken over, else the oldest stream is waited for); a CAPTURED launch owns its area like its spill region.
static unsigned long long* rc_totals_scratch(rc_scene* s, hipStream_t stream, bool capturing, size_t words) {
    if (words == 0) words = 1;
    if (capturing) {  // (inside an RcLaunchGuard that took a capture slot)
        auto& owned = s->capture_slots[s->cur_capture].scratch;
        owned.emplace_back(new DevBuf<unsigned long long>());
        owned.back()->reserve(words);
        return owned.back()->p;
    }
    size_t idx = 0;
    for (; idx < s->totals_scratch.size(); ++idx) if (s->totals_scratch[idx].stream == stream) break;
    if (idx == s->totals_scratch.size()) {
        if (s->totals_scratch.size() < 16) {
            s->totals_scratch.emplace_back();
            s->totals_scratch.back().buf.reset(new DevBuf<unsigned long long>());
        } else {  // take over the area whose last launch is done (the entry's own event: never the caller's stream handle), else wait for the oldest's
            size_t victim = s->totals_scratch.size();
            for (size_t i = 0; i < s->totals_scratch.size() && victim == s->totals_scratch.size(); ++i)
                if (s->totals_scratch[i].last.idle()) victim = i;
            if (victim == s->totals_scratch.size()) { victim = 0; s->totals_scratch[0].last.wait(); }
            std::rotate(s->totals_scratch.begin() + victim, s->totals_scratch.begin() + victim + 1, s->totals_scratch.end());
        }
        idx = s->totals_scratch.size() - 1;
        s->totals_scratch[idx].stream = stream;
    }
    s->totals_scratch[idx].buf->reserve(words);
    s->cur_scratch = (int)idx;
    return s->totals_scratch[idx].buf->p;
}

// Totals of rays [ray_begin, ray_end) of the sources with flat indices [src_begin, src_end), ACCUMULATED into d_received / d_emitted
// (n_prims u64 each, device; either may be nullptr).
void rc_launch_vf_totals(rc_scene* s, uint32_t rays_per_triangle, uint64_t seed, uint32_t src_begin, uint32_t src_end, uint32_t ray_begin,
                         uint32_t ray_end, unsigned long long* d_received, unsigned long long* d_emitted, hipStream_t stream) {
    if (src_end > s->n_flat_prims) src_end = s->n_flat_prims;
    if (ray_end > rays_per_triangle) ray_end = rays_per_triangle;
    if (src_begin >= src_end || ray_begin >= ray_end || (!d_received && !d_emitted)) return;
    check_buffer_range(s);
    const uint64_t total = (uint64_t)(src_end - src_begin) * (ray_end - ray_begin);
    RcLaunchGuard launch(s, stream);
    const bool partial = rc_partial_driver_ok(s);
    const bool lds = partial || rc_lds_driver_ok(s);
    const uint32_t bs = lds ? (uint32_t)kMidBlock : (uint32_t)kBlock;
    const uint32_t blocks = lds ? rc_lds_driver_blocks(s, total) : rc_persistent_blocks(s, total);
    SceneView v = rc_scene_view(s, blocks * bs);
    PersistArgs p = rc_persist_args(s, total, blocks * bs);
    const uint32_t np = s->n_flat_prims, n8 = (np + 7u) / 8u;
    unsigned long long* scratch = nullptr;
    if (d_received) {  // the private copies of the hot vector (see ViewFactorTotalsSink): one scratch area per stream, launches on one stream are ordered
        scratch = rc_totals_scratch(s, stream, launch.capturing, (size_t)kTotalsCopies * 8u * n8);
        RC_HIP(hipMemsetAsync(scratch, 0, sizeof(unsigned long long) * kTotalsCopies * 8u * n8, stream));
    }
    launch.start();
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32), n_ray = ray_end - ray_begin;
    if (partial) {
        rc_partial_driver_args(s, p);
        if (!s->lds_attr_set[11]) {
            RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_vf_totals_partial), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPartialLdsBytes));
            s->lds_attr_set[11] = true;
        }
        hipLaunchKernelGGL(k_vf_totals_partial, dim3(blocks), dim3(kMidBlock), kPartialLdsBytes, stream, v, p, k0, k1, src_begin, ray_begin, n_ray, scratch, n8, d_emitted);
    } else if (lds) {
        rc_lds_driver_args(s, p);
        if (!s->lds_attr_set[10]) {
            RC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_vf_totals_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMidLdsBytes));
            s->lds_attr_set[10] = true;
        }
        hipLaunchKernelGGL(k_vf_totals_lds, dim3(blocks), dim3(kMidBlock), kMidLdsBytes, stream, v, p, k0, k1, src_begin, ray_begin, n_ray, scratch, n8, d_emitted);
    } else
    hipLaunchKernelGGL(k_vf_totals, dim3(blocks), dim3(kBlock), 0, stream, v, p, k0, k1, src_begin, ray_begin, n_ray, scratch, n8, d_emitted);
    if (d_received) hipLaunchKernelGGL(k_vf_totals_fold, dim3((np + 255) / 256), dim3(256), 0, stream, scratch, n8, np, d_received);
    launch.finish();
}

// Kernels outside RcLaunchGuard that read scene memory on a caller's stream (the wavefront stages): remember the stream's latest such
// launch.  rc_scene_destroy waits for these events and for the launch slots' -- not for the whole device, which HIP refuses while any
// stream of the process is being captured.
void rc_note_stage_launch(rc_scene* s, hipStream_t stream) {
    if (stream == s->stream) return;  // the scene's own streams are synchronised at destruction
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing(stream, &st) == hipSuccess && st == hipStreamCaptureStatusActive) return;  // (a replay in flight is the caller's to wait for)
    std::lock_guard<std::mutex> lk(s->stage_mu);
    auto it = s->stage_events.find(stream);
    if (it == s->stage_events.end()) {
        if (s->stage_events.size() >= 64) {  // a caller that keeps making streams: retire one whose work is done
            for (auto jt = s->stage_events.begin(); jt != s->stage_events.end(); ++jt)
                if (hipEventQuery(jt->second) == hipSuccess) { (void)hipEventDestroy(jt->second); s->stage_events.erase(jt); break; }
            (void)hipGetLastError();
            if (s->stage_events.size() >= 64) { auto jt = s->stage_events.begin(); (void)hipEventSynchronize(jt->second); (void)hipEventDestroy(jt->second); s->stage_events.erase(jt); }
        }
        hipEvent_t ev = nullptr;
        RC_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        it = s->stage_events.emplace(stream, ev).first;
    }
    RC_HIP(hipEventRecord(it->second, stream));
}

void rc_launch_view_factor_rays(rc_scene* s, uint64_t seed, uint32_t src, uint32_t ray_begin, uint32_t n_ray, RcRay* d_out, hipStream_t stream) {
    if (n_ray == 0) return;
    SceneView v = rc_scene_view_static(s);
    hipLaunchKernelGGL(k_view_factor_rays, dim3((n_ray + 255) / 256), dim3(256), 0, stream, v, (uint32_t)seed, (uint32_t)(seed >> 32), src, ray_begin, n_ray, d_out);
    RC_HIP(hipGetLastError());
    rc_note_stage_launch(s, stream);
}

void rc_launch_hit_points(rc_scene* s, const RcRay* d_rays, const RcHit* d_hits, uint64_t n, float* d_points, float* d_normals, hipStream_t stream) {
    if (n == 0) return;
    uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 255) / 256, (uint64_t)s->n_cus * 8);
    hipLaunchKernelGGL(k_hit_points, dim3(blocks), dim3(256), 0, stream, rc_scene_view_static(s), d_rays, d_hits, n, d_points, d_normals);
    RC_HIP(hipGetLastError());
    rc_note_stage_launch(s, stream);
}

void rc_launch_shadow_rays(rc_scene* s, const RcRay* d_rays, const RcHit* d_hits, uint64_t n, const float light[3], float bias, RcRay* d_out, hipStream_t stream) {
    if (n == 0) return;
    uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 255) / 256, (uint64_t)s->n_cus * 8);
    hipLaunchKernelGGL(k_shadow_rays, dim3(blocks), dim3(256), 0, stream, rc_scene_view_static(s), d_rays, d_hits, n, light[0], light[1], light[2], bias, d_out);
    RC_HIP(hipGetLastError());
    rc_note_stage_launch(s, stream);
}

void rc_launch_primary_rays(rc_scene* s, const float pos[3], const float right[3], const float up[3], const float forward[3], float half_width,
                            float half_height, uint32_t width, uint32_t height, uint32_t samples, uint64_t seed, int jitter, RcRay* d_out, hipStream_t stream) {
    const uint64_t n = (uint64_t)width * height * samples;
    if (n == 0) return;
    CameraParams c;
    for (int k = 0; k < 3; ++k) { c.pos[k] = pos[k]; c.right[k] = right[k]; c.up[k] = up[k]; c.forward[k] = forward[k]; }
    c.half_width = half_width; c.half_height = half_height;
    c.width = width; c.height = height; c.samples = samples; c.jitter = jitter ? 1u : 0u;
    c.k0 = (uint32_t)seed; c.k1 = (uint32_t)(seed >> 32);
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((n + 255) / 256, (uint64_t)s->n_cus * 16);
    hipLaunchKernelGGL(k_primary_rays, dim3(blocks), dim3(256), 0, stream, c, n, d_out);
    RC_HIP(hipGetLastError());
}

// Indices of the rays that hit, ascending, and their number: the compaction step between wavefront stages (the reference's
// demo renderer keeps dummy rays instead, docs/src/wavefront-renderer.jl:445, so every later stage runs over all slots).
void rc_launch_compact_hits(rc_scene* s, const RcHit* d_hits, uint64_t n, uint32_t* d_indices, uint32_t* d_count, hipStream_t stream) {
    if (n == 0) { RC_HIP(hipMemsetAsync(d_count, 0, 4, stream)); return; }
    if (n > 0x7FFFFFFFull) throw RcError(1, "rc_compact_hits: more than 2^31 - 1 rays");
    s->compact_flags.reserve(n);
    s->compact_pos.reserve(n);
    const uint32_t blocks = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL(k_hit_flags, dim3(blocks), dim3(256), 0, stream, d_hits, n, s->compact_flags.p);
    size_t tmp = 0;
    RC_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp, s->compact_flags.p, s->compact_pos.p, (int)n, stream));
    s->compact_tmp.reserve(tmp ? tmp : 1);
    RC_HIP(hipcub::DeviceScan::ExclusiveSum(s->compact_tmp.p, tmp, s->compact_flags.p, s->compact_pos.p, (int)n, stream));
    hipLaunchKernelGGL(k_scatter_hit_indices, dim3(blocks), dim3(256), 0, stream, s->compact_flags.p, s->compact_pos.p, n, d_indices, d_count);
    RC_HIP(hipGetLastError());
    rc_note_stage_launch(s, stream);
}
