// Dev tool: cost of hipHostRegister / pinned staging vs pageable hipMemcpy for a 134 MB buffer (the 4 M-ray batch).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t bytes = (size_t)4194304 * 32;
    char* h = (char*)aligned_alloc(4096, bytes);
    memset(h, 1, bytes);
    char* d; hipMalloc((void**)&d, bytes);
    hipStream_t st; hipStreamCreate(&st);
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now(); hipMemcpy(d, h, bytes, hipMemcpyHostToDevice); double t1 = now();
        hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost); double t2 = now();
        printf("pageable H2D %.2f ms (%.1f GB/s)  D2H %.2f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9, (t2 - t1) * 1e3, bytes / (t2 - t1) / 1e9);
    }
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now(); hipError_t e = hipHostRegister(h, bytes, hipHostRegisterDefault); double t1 = now();
        hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); double t2 = now();
        hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); double t3 = now();
        hipHostUnregister(h); double t4 = now();
        printf("register(%d) %.2f ms  H2D %.2f ms (%.1f GB/s)  D2H %.2f ms (%.1f GB/s)  unregister %.2f ms\n", (int)e, (t1 - t0) * 1e3, (t2 - t1) * 1e3,
               bytes / (t2 - t1) / 1e9, (t3 - t2) * 1e3, bytes / (t3 - t2) / 1e9, (t4 - t3) * 1e3);
    }
    char* p; hipHostMalloc((void**)&p, bytes, hipHostMallocDefault);
    for (int nt : {1, 2, 4, 8, 16}) {
        double best = 1e9;
        for (int rep = 0; rep < 3; ++rep) {
            double t0 = now();
            std::vector<std::thread> th;
            for (int k = 0; k < nt; ++k) th.emplace_back([&, k] { size_t a = bytes * k / nt, b = bytes * (k + 1) / nt; memcpy(p + a, h + a, b - a); });
            for (auto& x : th) x.join();
            best = std::min(best, now() - t0);
        }
        printf("memcpy pageable->pinned, %2d threads: %.2f ms (%.1f GB/s)\n", nt, best * 1e3, bytes / best / 1e9);
    }
    {
        double t0 = now(); hipMemcpyAsync(d, p, bytes, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); double t1 = now();
        hipMemcpyAsync(p, d, bytes, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); double t2 = now();
        printf("pinned H2D %.2f ms (%.1f GB/s)  D2H %.2f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9, (t2 - t1) * 1e3, bytes / (t2 - t1) / 1e9);
    }
    return 0;
}
