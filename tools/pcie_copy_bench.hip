// PCIe copy rates of the host-pointer block path: 16 MiB blocks, malloc'ed memory registered in place (hipHostRegister) vs
// hipHostMalloc, each direction alone and both at once on two streams.
//   hipcc -O2 tools/pcie_copy_bench.hip -o /tmp/pcie && /tmp/pcie
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
    const size_t n = 16u << 20;
    void *d0, *d1; CK(hipMalloc(&d0, n)); CK(hipMalloc(&d1, n));
    hipStream_t s0, s1; CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
    for (int mode = 0; mode < 2; mode++) {
        void *h0, *h1;
        if (mode == 0) { h0 = aligned_alloc(4096, n); h1 = aligned_alloc(4096, n); memset(h0, 1, n); memset(h1, 2, n);
                         CK(hipHostRegister(h0, n, hipHostRegisterDefault)); CK(hipHostRegister(h1, n, hipHostRegisterDefault)); }
        else { CK(hipHostMalloc(&h0, n, hipHostMallocDefault)); CK(hipHostMalloc(&h1, n, hipHostMallocDefault)); memset(h0, 1, n); memset(h1, 2, n); }
        for (int what = 0; what < 3; what++) {
            const int reps = 50;
            for (int w = 0; w < 5; w++) { CK(hipMemcpyAsync(d0, h0, n, hipMemcpyHostToDevice, s0)); CK(hipMemcpyAsync(h1, d1, n, hipMemcpyDeviceToHost, s1)); }
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; r++) {
                if (what != 1) CK(hipMemcpyAsync(d0, h0, n, hipMemcpyHostToDevice, s0));
                if (what != 0) CK(hipMemcpyAsync(h1, d1, n, hipMemcpyDeviceToHost, s1));
            }
            CK(hipDeviceSynchronize());
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            printf("%-22s %-12s: %6.1f GB/s per direction (%.3f ms per 16 MiB block)\n", mode == 0 ? "registered in place" : "hipHostMalloc",
                   what == 0 ? "H2D alone" : what == 1 ? "D2H alone" : "both at once", n * reps / dt / 1e9, dt / reps * 1e3);
        }
    }
    return 0;
}
