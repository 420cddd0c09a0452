// How does this HIP runtime move PAGEABLE host memory in hipMemcpy[Async]?  Run with AMD_LOG_LEVEL=4 and look for
// "HSA Copy Using Pinned resource" (the runtime pins the caller's pages and the GPU / SDMA touches them) or
// "... Using Staging resource" (CPU copies through the runtime's own pinned buffer).   tools/pageable_copy_probe.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
int main(int argc, char **argv)
{
    const size_t sizes[] = {64u << 10, 1u << 20, 4u << 20, 16u << 20, 64u << 20, 200u << 20};
    hipStream_t s; (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (size_t bytes : sizes) {
        void *d = nullptr; (void)hipMalloc(&d, bytes);
        char *h = (char *)malloc(bytes); memset(h, 1, bytes);
        fprintf(stderr, "=== %zu bytes: async H2D\n", bytes);
        (void)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s); (void)hipStreamSynchronize(s);
        fprintf(stderr, "=== %zu bytes: async D2H\n", bytes);
        (void)hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s);
        fprintf(stderr, "=== %zu bytes: sync H2D\n", bytes);
        (void)hipMemcpy(d, h, bytes, hipMemcpyHostToDevice);
        fprintf(stderr, "=== %zu bytes: sync D2H\n", bytes);
        (void)hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost);
        free(h); (void)hipFree(d);
    }
    return 0;
}
