// Dev tool: cost of the work-claim pattern -- W waves each doing R returning atomicAdds on ONE counter (or on 8 counters 64 B / 4 KiB apart).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k32(unsigned int* c, int reps, int shards, int stride_words, unsigned long long* sink) {
    const int lane = threadIdx.x & 63;
    unsigned long long acc = 0;
    for (int r = 0; r < reps; ++r) {
        unsigned int v = 0;
        if (lane == 0) v = atomicAdd(c + (size_t)((blockIdx.x + r) % shards) * stride_words, 1u);
        acc += __shfl(v, 0);
    }
    if (acc == 0xdeadbeefull) sink[0] = acc;
}
__global__ void k_noret(float* c, int per_lane, int n_addr) {  // every lane: non-returning float atomics, addresses spread over n_addr words
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int r = 0; r < per_lane; ++r) atomicAdd(c + ((t * 2654435761u + r) % (unsigned)n_addr), 1.0f);
}
__global__ void k(unsigned long long* c, int reps, int shards, int stride_words, unsigned long long* sink) {
    const int lane = threadIdx.x & 63;
    unsigned long long acc = 0;
    for (int r = 0; r < reps; ++r) {
        unsigned long long v = 0;
        if (lane == 0) v = atomicAdd(c + (size_t)((blockIdx.x + r) % shards) * stride_words, 128ull);
        acc += __shfl(v, 0);
    }
    if (acc == 0xdeadbeefull) sink[0] = acc;
}
int main() {
    unsigned long long *c, *sink;
    hipMalloc((void**)&c, 1 << 20); hipMalloc((void**)&sink, 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int shards : {1, 16}) for (int stride : {8, 32}) for (int reps : {5}) {
        if (shards == 1 && stride != 8) continue;
        float best = 1e9;
        for (int it = 0; it < 5; ++it) {
            hipMemset(c, 0, 1 << 20);
            hipEventRecord(a);
            hipLaunchKernelGGL(k, dim3(6144), dim3(64), 0, 0, c, reps, shards, stride, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("6144 waves x %d atomics, %d counter(s) %d B apart: %.1f us  (%.1f ns per atomic)\n", reps, shards, stride * 8, best * 1e3, best * 1e6 / (6144.0 * reps));
    }
    for (int n_addr : {1, 16, 1024, 1 << 20}) {
        float best = 1e9;
        for (int it = 0; it < 3; ++it) {
            hipMemset(c, 0, 1 << 20);
            hipEventRecord(a);
            hipLaunchKernelGGL(k_noret, dim3(4096), dim3(256), 0, 0, (float*)c, 1, n_addr > (1 << 18) ? (1 << 18) : n_addr);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("1 Mi non-returning f32 atomics over %d address(es): %.1f us (%.2f ns each)\n", n_addr, best * 1e3, best * 1e6 / (4096.0 * 256));
    }
    for (int shards : {1, 16}) {
        float best = 1e9;
        for (int it = 0; it < 5; ++it) {
            hipMemset(c, 0, 1 << 20);
            hipEventRecord(a);
            hipLaunchKernelGGL(k32, dim3(6144), dim3(64), 0, 0, (unsigned int*)c + 64, 5, shards, 64, sink);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
        }
        printf("u32: 6144 waves x 5 atomics, %d counter(s) 256 B apart: %.1f us (%.1f ns per atomic)\n", shards, best * 1e3, best * 1e6 / (6144.0 * 5));
    }
    return 0;
}
