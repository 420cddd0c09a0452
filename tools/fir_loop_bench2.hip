// Second micro-benchmark of the FIR inner loop: 16 dependent MFMAs per iteration (4 groups of 4),
// operands for the NEXT 16 fetched under them with as few LDS instructions as possible:
// x rows read as adjacent pairs (one ds_read2_b64 per row per two groups), h with stride-4 pairs.
//   MODE 0 VGPR accumulator (builtin), MODE 1 AGPR accumulator (inline asm)
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 tools/fir_loop_bench2.hip -o /tmp/flb2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mfma_agpr(v4f64 &acc, double a, double b)
{
    asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, int ngroups, int row, const double *gh)
{
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 9000; i += 256) lds[i] = 1.0 + i * 1e-6;
    __syncthreads();
    const int lane = threadIdx.x & 63, ij = lane & 15, kk = lane >> 4;
    const double *hp = (MODE >= 2 ? gh + (blockIdx.x & 255) * 4352 : lds) + 1 + kk + ij;   // MODE>=2: taps straight from global (L1/L2)
    const double *xp[4];
    for (int s = 0; s < 4; s++) xp[s] = lds + 4300 + ((15 - kk - 4 * s) & 15) * row + 300 + ij;
    v4f64 acc = {0, 0, 0, 0};
    double ha[16], xa[16], hb[16], xb[16];               // operand sets for 4 groups each
    auto load = [&](double *h, double *x, int g) {       // groups g .. g+3
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int s = 0; s < 4; s++) { h[4 * q + s] = hp[16 * (g + q) + 4 * s]; x[4 * q + s] = xp[s][-(g + q)]; }
    };
    auto fma16 = [&](const double *h, const double *x) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (MODE == 1 || MODE == 3) mfma_agpr(acc, h[u], x[u]);
            else acc = __builtin_amdgcn_mfma_f64_16x16x4f64(h[u], x[u], acc, 0, 0, 0);
        }
    };
    load(ha, xa, 0);
    for (int rep = 0; rep < 32; rep++)
        for (int g = 0; g < ngroups; g += 8) {
            load(hb, xb, (g + 4) & 127);
            __builtin_amdgcn_sched_barrier(0);
            fma16(ha, xa);
            __builtin_amdgcn_sched_barrier(0);
            load(ha, xa, (g + 8) & 127);
            __builtin_amdgcn_sched_barrier(0);
            fma16(hb, xb);
            __builtin_amdgcn_sched_barrier(0);
        }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int MODE>
void run(const char *name, int wps, double *out)
{
    const int ngroups = 128, reps = 5;
    int blocks = 256 * wps;
    size_t lds = 9000 * 8;
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, out, ngroups, 336, out + 256 * 1024 * 8);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, out, ngroups, 336, out + 256 * 1024 * 8);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double t = ms * 1e-3 / reps, nm = 32.0 * ngroups * 4;
    printf("%-44s waves/SIMD=%d : %8.2f us/launch  %6.1f cycles/MFMA/SIMD @2.4GHz  %6.2f TFLOP/s\n", name, wps, t * 1e6,
           t * 2.4e9 / (nm * wps), blocks * 4 * nm * 2048.0 / t / 1e12);
}

int main()
{
    double *out; hipMalloc(&out, 256 * 1024 * 8 * 8 * 2); hipMemset(out, 0, 256 * 1024 * 8 * 8 * 2);
    for (int wps : {1, 2, 3}) {
        run<0>("16-deep sets, paired reads, VGPR acc", wps, out);
        run<1>("16-deep sets, paired reads, AGPR acc", wps, out);
        run<2>("taps from global, window from LDS, VGPR acc", wps, out);
        run<3>("taps from global, window from LDS, AGPR acc", wps, out);
    }
    return 0;
}
