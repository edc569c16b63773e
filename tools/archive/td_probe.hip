// Dev microbenchmark: cost of wave-level 16-byte gathers from a cache-resident table as a function of
// the number of active lanes, and the same through LDS.  Informs the traversal kernel's fetch design.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int LOADS>
__global__ __launch_bounds__(256) void k_gather(const float4* table, unsigned mask, int active, int iters, float* out) {
    const int lane = threadIdx.x & 63;
    unsigned idx = (blockIdx.x * 256 + threadIdx.x) * 2654435761u;
    float acc = 0.f;
    if (lane < active) {
        for (int i = 0; i < iters; ++i) {
            idx = idx * 1664525u + 1013904223u;
            const float4* p = table + ((idx >> 8) & mask) * 4;   // 64-byte aligned record
#pragma unroll
            for (int k = 0; k < LOADS; ++k) { float4 v = p[k]; acc += v.x + v.w; }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// quad-cooperative pattern: lane k of each quad loads chunk k of the quad's j-th record (j = 0..3), so every
// instruction reads one contiguous 64-byte record per quad instead of four scattered 16-byte chunks.
__global__ __launch_bounds__(256) void k_gather_quad(const float4* table, unsigned mask, int iters, float* out) {
    const int lane = threadIdx.x & 63, k = lane & 3, quad = (blockIdx.x * 256 + threadIdx.x) >> 2;
    unsigned idx = quad * 2654435761u;
    float acc = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            idx = idx * 1664525u + 1013904223u;   // same value in the 4 lanes of a quad
            float4 v = table[((idx >> 8) & mask) * 4 + k];
            acc += v.x + v.w;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(1024) void k_gather_lds(const float4* table, unsigned mask, int active, int iters, float* out) {
    extern __shared__ float4 cache[];
    for (unsigned i = threadIdx.x; i < (mask + 1) * 4; i += 1024) cache[i] = table[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned idx = (blockIdx.x * 1024 + threadIdx.x) * 2654435761u;
    float acc = 0.f;
    if (lane < active) {
        for (int i = 0; i < iters; ++i) {
            idx = idx * 1664525u + 1013904223u;
            const float4* p = cache + ((idx >> 8) & mask) * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) { float4 v = p[k]; acc += v.x + v.w; }
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

// LDS with the records split into planes: plane k holds chunk k of every record, so lanes reading chunk k of different
// records hit addresses 16 (or 8) bytes apart per record index instead of 64: 16 (32) distinct bank groups instead of 4.
template <int CHUNK_FLOATS>
__global__ __launch_bounds__(1024) void k_gather_lds_planes(const float4* table, unsigned mask, int active, int iters, float* out) {
    extern __shared__ float4 cache[];
    const unsigned n = mask + 1;
    float* cf = reinterpret_cast<float*>(cache);
    const float* tf = reinterpret_cast<const float*>(table);
    constexpr int PLANES = 16 / CHUNK_FLOATS;
    for (unsigned i = threadIdx.x; i < n * 16; i += 1024) {
        const unsigned rec = i / 16, w = i % 16, plane = w / CHUNK_FLOATS, e = w % CHUNK_FLOATS;
        cf[(plane * n + rec) * CHUNK_FLOATS + e] = tf[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned idx = (blockIdx.x * 1024 + threadIdx.x) * 2654435761u;
    float acc = 0.f;
    if (lane < active) {
        for (int i = 0; i < iters; ++i) {
            idx = idx * 1664525u + 1013904223u;
            const unsigned rec = (idx >> 8) & mask;
            if (CHUNK_FLOATS == 4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) { float4 v = cache[k * n + rec]; acc += v.x + v.w; }
            } else {
                const float2* c2 = reinterpret_cast<const float2*>(cache);
#pragma unroll
                for (int k = 0; k < PLANES; ++k) { float2 v = c2[k * n + rec]; acc += v.x + v.y; }
            }
        }
    }
    out[blockIdx.x * 1024 + threadIdx.x] = acc;
}

int main() {
    const int iters = 2000;
    float4* table; float* out;
    CK(hipMalloc(&table, 64u << 20)); CK(hipMalloc(&out, 4u << 20));
    CK(hipMemset(table, 0, 64u << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned masks[] = {255, 8191, 262143};   // 16 KB (L1), 512 KB (L2), 16 MB (MALL/L2 mix)
    for (unsigned mask : masks) {
        for (int active : {64, 48, 32, 16, 8, 4}) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_gather<4>, dim3(256 * 6), dim3(256), 0, 0, table, mask, active, iters, out);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            double waves = 256.0 * 6 * 4, insts = waves * iters * 4;
            // cycles of one CU's TA/TD per wave-level dwordx4 instruction, assuming 2.1 GHz and 24 waves sharing one CU
            printf("global table %6u KB active %2d: %.3f ms  -> %.1f clk/CU per load instr, %.1f GB/s useful\n", (mask + 1) / 16, active, ms,
                   ms * 1e-3 * 2.1e9 * 256 / insts, waves * iters * active * 64.0 / (ms * 1e-3) / 1e9);
        }
    }
    for (unsigned mask : masks) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_gather_quad, dim3(256 * 6), dim3(256), 0, 0, table, mask, iters, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double waves = 256.0 * 6 * 4, insts = waves * iters * 4;
        printf("quad-cooperative table %6u KB: %.3f ms -> %.1f clk/CU per load instr, %.1f GB/s useful\n", (mask + 1) / 16, ms,
               ms * 1e-3 * 2.1e9 * 256 / insts, waves * iters * 64 * 64.0 / (ms * 1e-3) / 1e9);
    }
    for (int active : {64, 32, 16, 8}) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_gather_lds, dim3(256), dim3(1024), 64 * 1024, 0, table, 1023u, active, iters, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double waves = 256.0 * 16, insts = waves * iters * 4;
        printf("LDS 64 KB active %2d: %.3f ms -> %.1f clk/CU per ds_read_b128, %.1f GB/s useful\n", active, ms, ms * 1e-3 * 2.1e9 * 256 / insts,
               waves * iters * active * 64.0 / (ms * 1e-3) / 1e9);
    }
    for (int active : {64, 48, 40, 32, 16, 8}) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_gather_lds_planes<4>, dim3(256), dim3(1024), 64 * 1024, 0, table, 1023u, active, iters, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double waves = 256.0 * 16, insts = waves * iters * 4;
        printf("LDS float4 planes active %2d: %.3f ms -> %.1f clk/CU per ds_read_b128 (%.1f per 64-B record)\n", active, ms, ms * 1e-3 * 2.1e9 * 256 / insts,
               ms * 1e-3 * 2.1e9 * 256 / insts * 4);
    }
    for (int active : {64, 40, 16}) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_gather_lds_planes<2>, dim3(256), dim3(1024), 64 * 1024, 0, table, 1023u, active, iters, out);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        }
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double waves = 256.0 * 16, insts = waves * iters * 8;
        printf("LDS float2 planes active %2d: %.3f ms -> %.1f clk/CU per ds_read_b64 (%.1f per 64-B record)\n", active, ms, ms * 1e-3 * 2.1e9 * 256 / insts,
               ms * 1e-3 * 2.1e9 * 256 / insts * 8);
    }
    return 0;
}
