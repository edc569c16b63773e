// Dev tool: calibrates rocprofv3's FETCH_SIZE for the trace kernels' access pattern -- random 64-byte records read with four
// 16-byte loads per lane from a table far larger than L2 + Infinity Cache -- against a known byte count.  MI355X_MICROARCH.md says
// FETCH_SIZE reports half of a wide coalesced stream on gfx950 and calls other patterns uncalibrated; this is that calibration.
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o /tmp/fetch_calib;  rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// mode 0: each lane reads one whole random 64-byte record (4 x dwordx4); mode 1: coalesced stream of 16 B per lane
__global__ __launch_bounds__(256) void k_gather64(const float4* table, unsigned long long n_records, int per_lane, float* out) {
    unsigned long long idx = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345ull;
    float acc = 0.f;
    for (int i = 0; i < per_lane; ++i) {
        idx = idx * 6364136223846793005ull + 1442695040888963407ull;
        const float4* p = table + ((idx >> 20) % n_records) * 4;
        float4 a = p[0], b = p[1], c = p[2], d = p[3];
        acc += a.x + b.y + c.z + d.w;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
// the same gather with eight records in flight per lane: the achievable rate of random 64-byte requests (the HBM-regime ceiling)
__global__ __launch_bounds__(256) void k_gather64_deep(const float4* table, unsigned long long n_records, int per_lane, float* out) {
    unsigned long long idx = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 777ull;
    float acc = 0.f;
    for (int i = 0; i < per_lane; i += 8) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            idx = idx * 6364136223846793005ull + 1442695040888963407ull;
            v[k] = table[((idx >> 20) % n_records) * 4 + (k & 3)];   // one 16-byte piece of eight different records: eight 64-byte requests
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k].x + v[k].w;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_stream(const float4* table, unsigned long long n_vec, float* out) {
    float acc = 0.f;
    for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < n_vec; i += (unsigned long long)gridDim.x * 256ull) { float4 a = table[i]; acc += a.x + a.w; }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    const unsigned long long bytes = 16ull << 30;  // 16 GiB table
    float4* table; float* out;
    CK(hipMalloc(&table, bytes));
    CK(hipMemset(table, 0, bytes));
    const int blocks = 256 * 8 * 4, per_lane = 16;
    CK(hipMalloc(&out, sizeof(float) * blocks * 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_gather64, dim3(blocks), dim3(256), 0, 0, table, bytes / 64, per_lane, out);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double b = (double)blocks * 256 * per_lane * 64;
        printf("k_gather64: %.0f bytes requested (random 64-byte records), %.3f ms, %.1f GB/s useful\n", b, ms, b / ms / 1e6);
    }
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_gather64_deep, dim3(blocks), dim3(256), 0, 0, table, bytes / 64, 64, out);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double req = (double)blocks * 256 * 64;
        printf("k_gather64_deep: %.0f random 64-byte requests (16 bytes used of each), %.3f ms, %.2f G requests/s = %.1f GB/s at 64 B per request\n", req, ms, req / ms / 1e6, req * 64 / ms / 1e6);
    }
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_stream, dim3(256 * 16), dim3(256), 0, 0, table, (4ull << 30) / 16, out);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("k_stream: %.0f bytes (coalesced 16 B per lane), %.3f ms, %.1f GB/s\n", (double)(4ull << 30), ms, (double)(4ull << 30) / ms / 1e6);
    }
    return 0;
}
