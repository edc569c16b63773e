// node_fetch_probe.hip -- what does the texture path charge for the ways a lane can fetch an interior node?  (round 5: sign-resolved slab test)
// Every lane walks a pseudo-random sequence of node records of an array that is L2-resident (0.5 MB, C3's BLAS) or not (12.8 / 25.6 MB,
// C2's), 6 waves per SIMD, 2 x 768 threads per CU like trace kernel 5, a few VALU per visit.  Patterns:
//   0  r4's record: 64-byte stride, 3 x dwordx4 + 1 x dwordx2, all 16-byte aligned
//   1  the ring record at 128-byte stride: 3 x dwordx4 at +0 / +8 (random per lane and axis) + dwordx2
//   2  pattern 1 with every window at +0 (aligned): the price of the stride alone
//   3  pattern 1 at 96-byte stride
//   4  64-byte stride, 7 x dwordx2 (near / far pairs fetched separately, no ring)
//   5  pattern 1 with every window at +8 (all misaligned)
//   6  64-byte record, 3 x dwordx4 aligned + dwordx2, at 128-byte stride (r4's loads, doubled footprint)
// hipcc --offload-arch=gfx950 -O3 -o node_fetch_probe node_fetch_probe.hip && ./node_fetch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u4v __attribute__((ext_vector_type(4)));
typedef unsigned int u2v __attribute__((ext_vector_type(2)));
__device__ inline __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000); }

template <int PAT>
__global__ __launch_bounds__(768, 2) void k_probe(const unsigned char* arr, uint32_t n_nodes, uint32_t iters, uint32_t* out) {
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(arr, 0xFFFFFFF0u);
    uint32_t x = (blockIdx.x * 768u + threadIdx.x) * 2654435761u + 12345u;
    uint32_t acc = 0;
    const uint32_t gx = (x >> 7) & 8u, gy = (x >> 9) & 8u, gz = (x >> 11) & 8u;  // per-lane "signs", fixed like a ray's
    for (uint32_t it = 0; it < iters; ++it) {
        x = x * 1664525u + 1013904223u;
        const uint32_t node = (x >> 8) % n_nodes;
        if (PAT == 0 || PAT == 6) {
            const uint32_t off = node * (PAT == 0 ? 64u : 128u);
            u4v a = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0), b = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 16, 0), c = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 32, 0);
            u2v d = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 48, 0);
            acc += a.x ^ a.w ^ b.y ^ b.z ^ c.x ^ c.w ^ d.x ^ d.y;
        } else if (PAT == 4) {
            const uint32_t off = node * 64u;
            const uint32_t ax = off + gx, ay = off + 16u + gy, az = off + 32u + gz;
            u2v a = __builtin_amdgcn_raw_buffer_load_b64(rs, ax, 0, 0), b = __builtin_amdgcn_raw_buffer_load_b64(rs, ax ^ 8u, 0, 0);
            u2v c = __builtin_amdgcn_raw_buffer_load_b64(rs, ay, 0, 0), d = __builtin_amdgcn_raw_buffer_load_b64(rs, ay ^ 8u, 0, 0);
            u2v e = __builtin_amdgcn_raw_buffer_load_b64(rs, az, 0, 0), f = __builtin_amdgcn_raw_buffer_load_b64(rs, az ^ 8u, 0, 0);
            u2v g = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 48, 0);
            acc += a.x ^ b.y ^ c.x ^ d.y ^ e.x ^ f.y ^ g.x ^ g.y;
        } else {
            const uint32_t stride = PAT == 3 ? 96u : 128u;
            const uint32_t off = node * stride;
            const uint32_t sx = PAT == 2 ? 0u : (PAT == 5 ? 8u : gx), sy = PAT == 2 ? 0u : (PAT == 5 ? 8u : gy), sz = PAT == 2 ? 0u : (PAT == 5 ? 8u : gz);
            u4v a = __builtin_amdgcn_raw_buffer_load_b128(rs, off + sx, 0, 0);
            u2v d = __builtin_amdgcn_raw_buffer_load_b64(rs, off + sx, 24, 0);
            u4v b = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 40u + sy, 0, 0), c = __builtin_amdgcn_raw_buffer_load_b128(rs, off + 64u + sz, 0, 0);
            acc += a.x ^ a.w ^ b.y ^ b.z ^ c.x ^ c.w ^ d.x ^ d.y;
        }
    }
    out[blockIdx.x * 768u + threadIdx.x] = acc;
}

template <int PAT>
static float run(const unsigned char* arr, uint32_t n_nodes, uint32_t iters, uint32_t* out, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_probe<PAT>, dim3(blocks), dim3(768), 0, 0, arr, n_nodes, iters / 4, out);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_probe<PAT>, dim3(blocks), dim3(768), 0, 0, arr, n_nodes, iters, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, blocks = cus * 2;
    const size_t cap = 64u << 20;
    unsigned char* arr; uint32_t* out;
    hipMalloc(&arr, cap); hipMemset(arr, 1, cap);
    hipMalloc(&out, (size_t)blocks * 768 * 4);
    const uint32_t iters = 2000;
    printf("%d CUs, %d workgroups of 768; ns per wave-visit per CU (one visit = one node fetched by all 64 lanes of a wave)\n", cus, blocks);
    const char* names[7] = {"0 r4 record, stride 64, 3 x4 + x2 aligned", "1 ring, stride 128, windows +0/+8 per lane", "2 ring, stride 128, windows +0", "3 ring, stride 96, windows +0/+8",
                            "4 stride 64, 7 x dwordx2 (near/far apart)", "5 ring, stride 128, windows +8", "6 r4 loads at stride 128"};
    for (uint32_t n_nodes : {8191u, 100000u, 200000u, 400000u}) {
        printf("nodes %u:\n", n_nodes);
        for (int p = 0; p < 7; ++p) {
            float ms = 0;
            switch (p) {
                case 0: ms = run<0>(arr, n_nodes, iters, out, blocks); break;
                case 1: ms = run<1>(arr, n_nodes, iters, out, blocks); break;
                case 2: ms = run<2>(arr, n_nodes, iters, out, blocks); break;
                case 3: ms = run<3>(arr, n_nodes, iters, out, blocks); break;
                case 4: ms = run<4>(arr, n_nodes, iters, out, blocks); break;
                case 5: ms = run<5>(arr, n_nodes, iters, out, blocks); break;
                case 6: ms = run<6>(arr, n_nodes, iters, out, blocks); break;
            }
            const double visits_per_cu = (double)iters * 24.0;  // 24 waves per CU
            printf("  %-48s %8.3f ms   %6.1f ns per wave-visit per CU  (%5.1f cycles at 2.4 GHz)\n", names[p], ms, ms * 1e6 / visits_per_cu, ms * 1e6 / visits_per_cu * 2.4);
        }
    }
    return 0;
}
