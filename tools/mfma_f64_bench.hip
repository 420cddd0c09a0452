// Micro-benchmark: what v_mfma_f64_16x16x4_f64 sustains on gfx950 as a function of independent
// accumulators per wave and waves per SIMD (no memory traffic).  Build on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 tools/mfma_f64_bench.hip -o /tmp/mfma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a0, double b0)
{
    v4f64 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = {0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// f64 VALU FMA for comparison
template <int NACC>
__global__ __launch_bounds__(256) void kf(double *out, int iters, double a0, double b0)
{
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = i;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; it++)
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int i = 0; i < NACC; i++) acc[i] = __builtin_fma(a, acc[i], b);
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
double timeit(F launch)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e-3;
}

int main()
{
    double *out; hipMalloc(&out, 256 * 4096 * 64 * sizeof(double));
    const int iters = 4000;
    printf("v_mfma_f64_16x16x4_f64: TFLOP/s by (accumulators per wave, waves per SIMD)\n");
    for (int wps : {1, 2, 4}) {
        int blocks = 256 * wps;                 // 256-thread blocks = 1 wave per SIMD each
        auto run = [&](auto kern, int nacc) {
            double t = timeit([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 2.0); });
            double flops = (double)blocks * 4 * iters * 8 * nacc * 2048.0;
            printf("  acc=%d waves/SIMD=%d : %7.2f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", nacc, wps, flops / t / 1e12,
                   t * 2.4e9 / ((double)wps * iters * 8 * nacc));
        };
        run(k<1>, 1); run(k<2>, 2); run(k<4>, 4);
    }
    printf("v_fma_f64 (VALU): TFLOP/s\n");
    for (int wps : {1, 2, 4}) {
        int blocks = 256 * wps;
        auto run = [&](auto kern, int nacc) {
            double t = timeit([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, iters * 4, 1.0, 2.0); });
            double flops = (double)blocks * 256 * iters * 4.0 * 8 * nacc * 2;
            printf("  chains=%d waves/SIMD=%d : %7.2f TFLOP/s\n", nacc, wps, flops / t / 1e12);
        };
        run(kf<1>, 1); run(kf<4>, 4); run(kf<8>, 8);
    }
    return 0;
}
