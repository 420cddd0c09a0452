// Micro-benchmark of the FIR inner loop shape: 4 dependent v_mfma_f64_16x16x4_f64 per group with the
// next group's 8 f64 operands read from LDS under them.  MODE 0: MFMA only; 1: LDS reads feed the MFMAs
// (as in fir_mfma); 2: LDS reads issued but MFMAs use loop-invariant operands; 3: like 1 but two
// independent accumulators (two column tiles per wave, A operand shared).
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 tools/fir_loop_bench.hip -o /tmp/fir_loop_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mfma_agpr(v4f64 &acc, double a, double b)
{
    asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, int ngroups, int row)
{
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 9000; i += 256) lds[i] = 1.0 + i * 1e-6;
    __syncthreads();
    const int lane = threadIdx.x & 63, ij = lane & 15, kk = lane >> 4;
    const double *hp = lds + 1 + kk + ij;
    const double *xp[4];
    for (int s = 0; s < 4; s++) xp[s] = lds + 4300 + ((15 - kk - 4 * s) & 15) * row + 300 + ij;
    v4f64 acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
    double h0[4], x0[4], h1[4], x1[4], z0[4];
    for (int s = 0; s < 4; s++) { h0[s] = hp[4 * s]; x0[s] = xp[s][0]; z0[s] = xp[s][1]; }
    double ca = 1.0 + lane, cb = 2.0 - lane;
    double sink = 0;
    for (int rep = 0; rep < 32; rep++)
    for (int g = 0; g < ngroups; g += 2) {
        if (MODE != 0 && MODE != 5) {
#pragma unroll
            for (int s = 0; s < 4; s++) { h1[s] = hp[16 * (g + 1) + 4 * s]; x1[s] = xp[s][-(g + 1)]; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; s++) {
            if (MODE == 4) mfma_agpr(acc, h0[s], x0[s]);
            else if (MODE == 5) mfma_agpr(acc, ca, cb);
            else if (MODE == 1 || MODE == 3) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(h0[s], x0[s], acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ca, cb, acc, 0, 0, 0);
            if (MODE == 3) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(h0[s], z0[s], acc2, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE != 0 && MODE != 5) {
#pragma unroll
            for (int s = 0; s < 4; s++) { h0[s] = hp[16 * ((g + 2) & 127) + 4 * s]; x0[s] = xp[s][-((g + 2) & 127)]; if (MODE == 3) z0[s] = xp[s][1 - ((g + 2) & 127)]; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; s++) {
            if (MODE == 4) mfma_agpr(acc, h1[s], x1[s]);
            else if (MODE == 5) mfma_agpr(acc, ca, cb);
            else if (MODE == 1 || MODE == 3) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(h1[s], x1[s], acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ca, cb, acc, 0, 0, 0);
            if (MODE == 3) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(h1[s], x1[s], acc2, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 2) for (int s = 0; s < 4; s++) asm volatile("" :: "v"(h0[s]), "v"(x0[s]), "v"(h1[s]), "v"(x1[s]));
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3] + acc2[0] + acc2[3] + sink;
}

template <int MODE>
void run(const char *name, int wps, double *out)
{
    const int ngroups = 128, reps = 5;                   // 32 passes * 128 groups * 4 = 16384 MFMA per launch
    int blocks = 256 * wps;
    size_t lds = 9000 * 8;
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, out, ngroups, 336);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, out, ngroups, 336);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double t = ms * 1e-3 / reps;
    double nm = 32.0 * ngroups * 4 * (MODE == 3 ? 2 : 1);
    printf("%-44s waves/SIMD=%d : %8.2f us/launch  %6.1f cycles/MFMA/SIMD @2.4GHz  %6.2f TFLOP/s\n", name, wps, t * 1e6,
           t * 2.4e9 / (nm * wps), blocks * 4 * nm * 2048.0 / t / 1e12);
}

int main()
{
    double *out; hipMalloc(&out, 256 * 1024 * 8 * 8);
    for (int wps : {1, 2}) {
        run<0>("mfma only", wps, out);
        run<1>("lds operands feed the mfma (fir_mfma shape)", wps, out);
        run<2>("lds reads issued, mfma operands constant", wps, out);
        run<3>("two accumulators, shared A operand", wps, out);
        run<5>("mfma only, accumulator in AGPRs", wps, out);
        run<4>("lds operands, accumulator in AGPRs", wps, out);
    }
    return 0;
}
