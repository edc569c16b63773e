// vf_host_probe.hip -- how fast can a row block of a view-factor matrix reach a caller's COLUMN-major host matrix?
//
// rc_view_factors returns a host N x N UInt32 matrix (src/kernels.jl:76).  A device row block [r0, r0 + C) of it, kept on the device as
// a column-major C x N block, maps to N pieces of C * 4 bytes, N * 4 bytes apart in the host matrix: a 2-D copy.  Measured here, for a
// 50 028-column matrix and blocks of 1 280 rows (256 MB), into pageable / registered host memory:
//   (a) hipHostRegister of the whole matrix (what a caller pays to pin it),
//   (b) hipMemcpy2DAsync device -> host,
//   (c) a copy kernel storing straight into the registered (device-mapped) matrix,
//   (d) a contiguous hipMemcpyAsync of the same bytes (the link's rate, for reference).
// Build: hipcc --offload-arch=gfx950 -O2 tools/vf_host_probe.hip -o tools/vf_host_probe
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// block: column-major C x N (element (r, c) at r + C * c); host: column-major, leading dimension ld, rows [r0, r0 + C)
__global__ void k_block_to_host(const uint32_t* __restrict__ block, uint32_t* __restrict__ host, uint32_t C, uint32_t N, uint64_t ld, uint32_t r0) {
    for (uint32_t c = blockIdx.x; c < N; c += gridDim.x) {
        const uint32_t* src = block + (size_t)c * C;
        uint32_t* dst = host + (size_t)c * ld + r0;
        for (uint32_t r = threadIdx.x; r < C; r += blockDim.x) dst[r] = src[r];
    }
}
__global__ void k_block_to_host4(const uint4* __restrict__ block, uint4* __restrict__ host, uint32_t C4, uint32_t N, uint64_t ld4, uint32_t r04) {
    for (uint32_t c = blockIdx.x; c < N; c += gridDim.x) {
        const uint4* src = block + (size_t)c * C4;
        uint4* dst = host + (size_t)c * ld4 + r04;
        for (uint32_t r = threadIdx.x; r < C4; r += blockDim.x) dst[r] = src[r];
    }
}

int main(int argc, char** argv) {
    const uint32_t N = argc > 1 ? atoi(argv[1]) : 50028, C = argc > 2 ? atoi(argv[2]) : 1280;
    const size_t total = (size_t)N * N * 4, blk = (size_t)C * N * 4;
    printf("matrix %u x %u u32 = %.2f GB, row block %u rows = %.1f MB\n", N, N, total / 1e9, C, blk / 1e6);
    uint32_t* host = (uint32_t*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (host == MAP_FAILED) { printf("mmap failed\n"); return 1; }
    double t0 = now();
    madvise(host, total, 23 /* MADV_POPULATE_WRITE */);
    printf("populate (MADV_POPULATE_WRITE) of the matrix: %.3f s\n", now() - t0);
    t0 = now();
    memset(host, 0, total);
    printf("memset of the populated matrix (1 thread): %.3f s = %.1f GB/s\n", now() - t0, total / (now() - t0) / 1e9);
    uint32_t* d_blk;
    CK(hipMalloc(&d_blk, blk));
    CK(hipMemset(d_blk, 1, blk));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int n_blocks = 8;
    // (b1) 2-D copy into pageable memory
    for (int rep = 0; rep < 2; ++rep) {
        t0 = now();
        for (int b = 0; b < n_blocks; ++b)
            CK(hipMemcpy2DAsync(host + (size_t)b * C, (size_t)N * 4, d_blk, (size_t)C * 4, (size_t)C * 4, N, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        printf("hipMemcpy2DAsync D2H, pageable, %d blocks: %.1f GB/s\n", n_blocks, n_blocks * blk / (now() - t0) / 1e9);
    }
    // (d1) contiguous D2H into pageable
    for (int rep = 0; rep < 2; ++rep) {
        t0 = now();
        for (int b = 0; b < n_blocks; ++b) CK(hipMemcpyAsync(host + (size_t)b * (blk / 4), d_blk, blk, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        printf("hipMemcpyAsync D2H contiguous, pageable, %d blocks: %.1f GB/s\n", n_blocks, n_blocks * blk / (now() - t0) / 1e9);
    }
    // (a) register
    t0 = now();
    CK(hipHostRegister(host, total, hipHostRegisterDefault));
    printf("hipHostRegister of %.2f GB: %.3f s\n", total / 1e9, now() - t0);
    uint32_t* host_dev = nullptr;
    CK(hipHostGetDevicePointer((void**)&host_dev, host, 0));
    for (int rep = 0; rep < 2; ++rep) {
        t0 = now();
        for (int b = 0; b < n_blocks; ++b)
            CK(hipMemcpy2DAsync(host + (size_t)b * C, (size_t)N * 4, d_blk, (size_t)C * 4, (size_t)C * 4, N, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        printf("hipMemcpy2DAsync D2H, registered, %d blocks: %.1f GB/s\n", n_blocks, n_blocks * blk / (now() - t0) / 1e9);
    }
    for (int rep = 0; rep < 2; ++rep) {
        t0 = now();
        for (int b = 0; b < n_blocks; ++b) CK(hipMemcpyAsync(host + (size_t)b * (blk / 4), d_blk, blk, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        printf("hipMemcpyAsync D2H contiguous, registered, %d blocks: %.1f GB/s\n", n_blocks, n_blocks * blk / (now() - t0) / 1e9);
    }
    for (int grid : {64, 256, 1024, 4096}) {
        for (int rep = 0; rep < 2; ++rep) {
            t0 = now();
            for (int b = 0; b < n_blocks; ++b) hipLaunchKernelGGL(k_block_to_host, dim3(grid), dim3(256), 0, st, d_blk, host_dev, C, N, (uint64_t)N, b * C);
            CK(hipStreamSynchronize(st));
            printf("copy kernel (dword stores, grid %d) into the mapped matrix, %d blocks: %.1f GB/s\n", grid, n_blocks, n_blocks * blk / (now() - t0) / 1e9);
        }
    }
    if (C % 4 == 0 && N % 4 == 0) {
        for (int grid : {64, 256, 1024, 4096}) {
            for (int rep = 0; rep < 2; ++rep) {
                t0 = now();
                for (int b = 0; b < n_blocks; ++b)
                    hipLaunchKernelGGL(k_block_to_host4, dim3(grid), dim3(256), 0, st, (const uint4*)d_blk, (uint4*)host_dev, C / 4, N, (uint64_t)N / 4, b * C / 4);
                CK(hipStreamSynchronize(st));
                printf("copy kernel (dwordx4 stores, grid %d) into the mapped matrix, %d blocks: %.1f GB/s\n", grid, n_blocks, n_blocks * blk / (now() - t0) / 1e9);
            }
        }
    }
    // spot check
    printf("host[0] = %08x host[%u] = %08x\n", host[0], C * n_blocks - 1, host[C * n_blocks - 1]);
    t0 = now();
    CK(hipHostUnregister(host));
    printf("hipHostUnregister: %.3f s\n", now() - t0);
    return 0;
}
