// Fourth micro-benchmark: two column tiles per wave (two independent accumulators sharing the taps
// operand), operand sets of NG groups, x rows read as adjacent pairs.  waves/SIMD 1..4.
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 tools/fir_loop_bench4.hip -o /tmp/flb4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int NG, int NACC>
__global__ __launch_bounds__(256) void k(double *out, int ngroups, int row)
{
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 4000; i += 256) lds[i] = 1.0 + i * 1e-6;
    __syncthreads();
    const int lane = threadIdx.x & 63, ij = lane & 15, kk = lane >> 4;
    const double *hp = lds + 1 + kk + ij;
    const double *xp[NACC][4];
    for (int t = 0; t < NACC; t++)
        for (int s = 0; s < 4; s++) xp[t][s] = lds + 1200 + ((15 - kk - 4 * s) & 15) * row + 100 + 16 * t + ij;
    v4f64 acc[NACC];
    for (int t = 0; t < NACC; t++) acc[t] = {0, 0, 0, 0};
    double ha[4 * NG], xa[NACC][4 * NG], hb[4 * NG], xb[NACC][4 * NG];
    auto load = [&](double *h, double (*x)[4 * NG], int g) {
#pragma unroll
        for (int q = 0; q < NG; q++)
#pragma unroll
            for (int s = 0; s < 4; s++) {
                h[4 * q + s] = hp[16 * (g + q) + 4 * s];
#pragma unroll
                for (int t = 0; t < NACC; t++) x[t][4 * q + s] = xp[t][s][-(g + q)];
            }
    };
    auto fma = [&](const double *h, double (*x)[4 * NG]) {
#pragma unroll
        for (int u = 0; u < 4 * NG; u++)
#pragma unroll
            for (int t = 0; t < NACC; t++) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(h[u], x[t][u], acc[t], 0, 0, 0);
    };
    load(ha, xa, 0);
    for (int rep = 0; rep < 64; rep++)
        for (int g = 0; g < ngroups; g += 2 * NG) {
            load(hb, xb, (g + NG) & 63);
            __builtin_amdgcn_sched_barrier(0);
            fma(ha, xa);
            __builtin_amdgcn_sched_barrier(0);
            load(ha, xa, (g + 2 * NG) & 63);
            __builtin_amdgcn_sched_barrier(0);
            fma(hb, xb);
            __builtin_amdgcn_sched_barrier(0);
        }
    double s = 0;
    for (int t = 0; t < NACC; t++) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NG, int NACC>
void run(const char *name, int wps, double *out)
{
    const int ngroups = 64, reps = 5;
    int blocks = 256 * wps;
    size_t lds = 4000 * 8;
    hipFuncSetAttribute((const void *)k<NG, NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k<NG, NACC>), dim3(blocks), dim3(256), lds, 0, out, ngroups, 144);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k<NG, NACC>), dim3(blocks), dim3(256), lds, 0, out, ngroups, 144);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double t = ms * 1e-3 / reps, nm = 64.0 * ngroups * 4 * NACC;
    printf("%-34s waves/SIMD=%d : %8.2f us/launch  %6.1f cycles/MFMA/SIMD @2.4GHz  %6.2f TFLOP/s\n", name, wps, t * 1e6,
           t * 2.4e9 / (nm * wps), blocks * 4 * nm * 2048.0 / t / 1e12);
}

int main()
{
    double *out; hipMalloc(&out, 256 * 1024 * 8 * 8);
    for (int wps : {1, 2, 3, 4}) {
        run<2, 1>("1 acc, sets of 2 groups", wps, out);
        run<4, 1>("1 acc, sets of 4 groups", wps, out);
        run<2, 2>("2 acc, sets of 2 groups", wps, out);
        if (wps <= 2) run<4, 2>("2 acc, sets of 4 groups", wps, out);
        if (wps <= 2) run<2, 4>("4 acc, sets of 2 groups", wps, out);
    }
    return 0;
}
