// Dev microbenchmark: VALU issue rate on gfx950 by instruction class, at the traversal kernel's occupancy (6 waves / SIMD).
// The trace kernels are bound by VALU issue (profiles/r02_*: SQ_ACTIVE_INST_VALU), so the ceiling of the roofline is
// "wave-instructions per second" of the mix the kernel issues.  Each kernel runs N dependent-free chains of one opcode.
// build: hipcc --offload-arch=gfx950 -O3 tools/valu_probe.hip -o tools/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kUnroll = 8;  // independent chains per lane

template <int OP>
__global__ __launch_bounds__(256, 6) void k_valu(float* out, int iters, float seed) {
    float a[kUnroll], b[kUnroll];
    v2f p[kUnroll], q[kUnroll];
    unsigned long long w[kUnroll];
    unsigned long long mask = 0x5555AAAA5555AAAAull + (unsigned long long)iters, mask2 = 0;
    const float negzero = __builtin_amdgcn_readfirstlane(iters) > 1 ? -0.0f : seed;
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) w[k] = threadIdx.x + k;
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) { a[k] = seed + k + threadIdx.x; b[k] = seed * 0.5f + k; p[k] = v2f{a[k], b[k]}; q[k] = v2f{b[k], a[k] * 0.25f}; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < kUnroll; ++k) {
            if (OP == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
            if (OP == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(q[k]));
            if (OP == 2) asm volatile("v_maximum3_f32 %0, %0, %1, %1" : "+v"(a[k]) : "v"(b[k]));
            if (OP == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[k]) : "v"(b[k]));
            if (OP == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[k]));
            if (OP == 5) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[k]) : "v"(b[k]));
            if (OP == 6) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[k]) : "v"(q[k]));
            if (OP == 7) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[k]), "v"(b[k]) : "vcc");
            if (OP == 8) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
            if (OP == 9) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a[k]) : "v"(b[k]));
            if (OP == 10) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[k]) : "v"(q[k]));
            if (OP == 11) asm volatile("v_minimum3_f32 %0, %0, %1, %1" : "+v"(a[k]) : "v"(b[k]));
            if (OP == 12) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b[k]), "s"(mask));
            if (OP == 13) asm volatile("v_mov_b32 %0, %1" : "+v"(a[k]) : "v"(b[k]));
            if (OP == 14) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
            if (OP == 15) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
            if (OP == 16) asm volatile("v_mul_f32_e64 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
            if (OP == 17) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(w[k]) : "v"(b[k]) : "vcc");
            if (OP == 18) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b[k]), "v"(p[k].y));
            if (OP == 19) asm volatile("v_swap_b32 %0, %1" : "+v"(a[k]), "+v"(b[k]));
            if (OP == 20) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
            if (OP == 21) asm volatile("v_maximum3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b[k]), "v"(p[k].y));
            if (OP == 22) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(mask2) : "v"(a[k]), "v"(b[k]));
            if (OP == 23) asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %0, %0, %2" : "+v"(a[k]) : "v"(b[k]), "v"(p[k].y));  // two instructions per count
            if (OP == 24) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b[k]), "v"(p[k].y));
            if (OP == 25) asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
            if (OP == 26) asm volatile("v_max_f32_e64 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
            if (OP == 27) asm volatile("v_min_f32_e64 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
            if (OP == 28) asm volatile("v_and_b32_e64 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
            if (OP == 29) asm volatile("v_mov_b32_e64 %0, %1" : "+v"(a[k]) : "v"(b[k]));
            if (OP == 30) asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
            if (OP == 31) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(a[k]) : "v"(b[k]), "v"(p[k].y));          // VOP2, destination not a source
            if (OP == 32) asm volatile("v_mul_f32_e64 %0, %1, %2" : "=v"(a[k]) : "v"(b[k]), "v"(p[k].y));
            if (OP == 33) asm volatile("v_maximum3_f32 %0, %1, %2, %2" : "=v"(a[k]) : "v"(b[k]), "v"(p[k].y));
            if (OP == 34) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p[k]) : "v"(q[k]), "v"(p[(k + 1) % kUnroll]));
            if (OP == 35) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2\n v_cndmask_b32_e64 %3, %1, %2, %0" : "=&s"(mask2), "+v"(a[k]), "+v"(b[k]), "=v"(p[k].x));  // cmp + select pair
            if (OP == 36) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %0, %1, vcc" : "+v"(a[k]), "+v"(b[k]), "=v"(p[k].x) : : "vcc");      // VOP2/VOPC forms through vcc
            if (OP == 37) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b[k]), "v"(p[k].y));
            if (OP == 38) asm volatile("v_sub_f32_e64 %0, %0, %1" : "+v"(a[k]) : "v"(b[k]));
            if (OP == 39) asm volatile("v_lshlrev_b32_e64 %0, 3, %0" : "+v"(a[k]));
            if (OP == 40) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b[k]), "s"(negzero));
        }
    }
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) acc += a[k] + p[k].x + p[k].y + (float)w[k] + b[k];
    acc += (float)mask2;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// Does a wave64 VALU instruction get cheaper when whole 16- or 32-lane groups are masked off?  (It would make lane compaction pay.)
__global__ __launch_bounds__(256, 6) void k_valu_masked(float* out, int iters, float seed, int active_lanes) {
    float a[kUnroll], b[kUnroll];
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) { a[k] = seed + k + threadIdx.x; b[k] = seed * 0.5f + k; }
    if ((int)(threadIdx.x & 63) < active_lanes) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < kUnroll; ++k) asm volatile("v_maximum3_f32 %0, %0, %1, %1" : "+v"(a[k]) : "v"(b[k]));
        }
    }
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) acc += a[k];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
static int run_masked(float* d_out, int n_cus, int active) {
    const int iters = 20000, blocks = n_cus * 6;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_valu_masked, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.5f, active);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("v_maximum3_f32 with lanes 0..%2d active: %.3f ns per wave-instruction per SIMD\n", active - 1, best * 1e6 / (6.0 * iters * kUnroll));
    return 0;
}

template <int OP>
static int run(const char* name, float* d_out, int n_cus) {
    const int iters = 20000, blocks = n_cus * 6;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_valu<OP>), dim3(blocks), dim3(256), 0, 0, d_out, 200, 1.5f);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_valu<OP>), dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.5f);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    // wave-instructions per SIMD: 6 waves x iters x kUnroll
    const double inst_per_simd = 6.0 * iters * kUnroll;
    const double ns_per_inst = best * 1e6 / inst_per_simd;
    printf("%-18s %8.3f ms  %.3f ns per wave-instruction per SIMD  => %.2f G wave-inst/s chip-wide (%d SIMDs)\n", name, best, ns_per_inst,
           n_cus * 4 / ns_per_inst, n_cus * 4);
    return 0;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, n_cus, prop.clockRate);
    float* d_out;
    CK(hipMalloc(&d_out, sizeof(float) * n_cus * 6 * 256));
    for (int active : {64, 48, 32, 16, 8, 1}) if (run_masked(d_out, n_cus, active)) return 1;
    if (run<0>("v_mul_f32", d_out, n_cus)) return 1;
    if (run<1>("v_pk_mul_f32", d_out, n_cus)) return 1;
    if (run<6>("v_pk_add_f32", d_out, n_cus)) return 1;
    if (run<10>("v_pk_fma_f32", d_out, n_cus)) return 1;
    if (run<5>("v_fma_f32", d_out, n_cus)) return 1;
    if (run<2>("v_maximum3_f32", d_out, n_cus)) return 1;
    if (run<11>("v_minimum3_f32", d_out, n_cus)) return 1;
    if (run<3>("v_cndmask_b32", d_out, n_cus)) return 1;
    if (run<7>("v_cmp_lt_f32", d_out, n_cus)) return 1;
    if (run<8>("v_add_u32", d_out, n_cus)) return 1;
    if (run<9>("v_lshl_add_u32", d_out, n_cus)) return 1;
    if (run<4>("v_rcp_f32", d_out, n_cus)) return 1;
    if (run<12>("v_cndmask_b32_e64 sgpr", d_out, n_cus)) return 1;
    if (run<13>("v_mov_b32", d_out, n_cus)) return 1;
    if (run<14>("v_max_f32", d_out, n_cus)) return 1;
    if (run<15>("v_add_f32", d_out, n_cus)) return 1;
    if (run<16>("v_mul_f32_e64", d_out, n_cus)) return 1;
    if (run<17>("v_mad_u64_u32", d_out, n_cus)) return 1;
    if (run<18>("v_fma_f32 3 regs", d_out, n_cus)) return 1;
    if (run<19>("v_swap_b32", d_out, n_cus)) return 1;
    if (run<20>("v_and_b32", d_out, n_cus)) return 1;
    if (run<21>("v_maximum3 3 regs", d_out, n_cus)) return 1;
    if (run<22>("v_cmp_lt_f32_e64 sgpr", d_out, n_cus)) return 1;
    if (run<23>("v_mul+v_add pair", d_out, n_cus)) return 1;
    if (run<24>("v_max3_f32", d_out, n_cus)) return 1;
    if (run<25>("v_add_f32_e64", d_out, n_cus)) return 1;
    if (run<26>("v_max_f32_e64", d_out, n_cus)) return 1;
    if (run<27>("v_min_f32_e64", d_out, n_cus)) return 1;
    if (run<28>("v_and_b32_e64", d_out, n_cus)) return 1;
    if (run<29>("v_mov_b32_e64", d_out, n_cus)) return 1;
    if (run<30>("v_add_u32_e64", d_out, n_cus)) return 1;
    if (run<31>("v_mul_f32 d!=s", d_out, n_cus)) return 1;
    if (run<32>("v_mul_f32_e64 d!=s", d_out, n_cus)) return 1;
    if (run<33>("v_maximum3 d!=s", d_out, n_cus)) return 1;
    if (run<34>("v_pk_mul_f32 d!=s", d_out, n_cus)) return 1;
    if (run<35>("cmp_e64+cndmask_e64", d_out, n_cus)) return 1;
    if (run<36>("cmp_vcc+cndmask_vcc", d_out, n_cus)) return 1;
    if (run<37>("v_med3_f32", d_out, n_cus)) return 1;
    if (run<38>("v_sub_f32_e64", d_out, n_cus)) return 1;
    if (run<39>("v_lshlrev_b32_e64", d_out, n_cus)) return 1;
    if (run<40>("v_fma_f32 x*y-0", d_out, n_cus)) return 1;
    return 0;
}
