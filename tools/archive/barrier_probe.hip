// Dev probe: does s_barrier count only the waves of a workgroup that are still running?  Waves leave at different times while the
// others keep meeting at barriers.  (Run under `timeout`: if exited waves were still counted this would never finish.)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(768) void k_probe(int* out, int rounds) {
    __shared__ int counter;
    if (threadIdx.x == 0) counter = 0;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    const int my_rounds = rounds * (wave + 1) / 12;   // wave 0 leaves first, wave 11 last
    int seen = 0;
    for (int r = 0; r < my_rounds; ++r) {
        if ((threadIdx.x & 63) == 0) atomicAdd(&counter, 1);
        __builtin_amdgcn_s_barrier();
        seen += counter;
        __builtin_amdgcn_s_barrier();
    }
    out[blockIdx.x * 768 + threadIdx.x] = seen;
}
int main() {
    int* d; CK(hipMalloc(&d, sizeof(int) * 768 * 512));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rounds : {12, 1200, 12000}) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_probe, dim3(512), dim3(768), 0, 0, d, rounds);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("rounds %d: finished in %.3f ms (%.1f ns per barrier pair for the longest wave)\n", rounds, ms, ms * 1e6 / rounds);
    }
    return 0;
}
