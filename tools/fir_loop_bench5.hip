// Fifth micro-benchmark: MFMA operands kept in registers and advanced with DPP row shifts instead of
// being re-read from LDS for every MFMA.
//   A (taps, Toeplitz): the operand of K-step s+1 is the operand of step s moved 4 lanes down within
//     each row of 16, the 4 vacated lanes filled from the next group's first operand  -> 1 LDS read / 4 MFMA
//   B (window): the operand of (group g+1, step s) is the operand of (g, s) moved one lane up, lane 0
//     of each row taking one new sample; one ds_read_b64 carries the new samples of 4 groups -> 1 read / 16 MFMA
// Variants: LDS (the fir_mfma shape: 2 reads per MFMA), DPP_A (A shifted, B from LDS), DPP_AB (both).
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 tools/fir_loop_bench5.hip -o /tmp/flb5
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ double dpp64(double old, double src)
{
    const long long o = __double_as_longlong(old), s = __double_as_longlong(src);
    const int lo = __builtin_amdgcn_update_dpp((int)o, (int)s, CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(s >> 32), CTRL, 0xF, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
#define ROW_SHL(n) (0x100 + (n))
#define ROW_SHR(n) (0x110 + (n))

template <int MODE>   // 0 = LDS both, 1 = DPP A, 2 = DPP A and B
__global__ __launch_bounds__(256) void k(double *out, int ngroups, int row)
{
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 4000; i += 256) lds[i] = 1.0 + i * 1e-6;
    __syncthreads();
    const int lane = threadIdx.x & 63, ij = lane & 15, kk = lane >> 4;
    const double *hp = lds + 1 + kk + ij;
    const double *xp[4];
    for (int s = 0; s < 4; s++) xp[s] = lds + 1200 + ((15 - kk - 4 * s) & 15) * row + 100 + ij;
    const double *lp = lds + 3000 + lane;
    v4f64 acc = {0, 0, 0, 0};
    double hcur = hp[0], hnext = hp[16];
    double B0 = xp[0][0], B1 = xp[1][0], B2 = xp[2][0], B3 = xp[3][0], L = lp[0];
    for (int rep = 0; rep < 64; rep++)
        for (int g = 0; g < ngroups; g += 4) {
            const double Lnext = (MODE == 2) ? lp[(g + 4) & 63] : 0.0;
#define GROUP(GG) { \
                const int gi = (g + GG) & 63; \
                const double hnn = hp[16 * ((gi + 2) & 63)]; \
                double A0, A1, A2, A3; \
                if (MODE >= 1) { \
                    A0 = hcur; \
                    A1 = dpp64<ROW_SHL(4)>(dpp64<ROW_SHR(12)>(hcur, hnext), hcur); \
                    A2 = dpp64<ROW_SHL(8)>(dpp64<ROW_SHR(8)>(hcur, hnext), hcur); \
                    A3 = dpp64<ROW_SHL(12)>(dpp64<ROW_SHR(4)>(hcur, hnext), hcur); \
                } else { \
                    A0 = hp[16 * gi]; A1 = hp[16 * gi + 4]; A2 = hp[16 * gi + 8]; A3 = hp[16 * gi + 12]; \
                } \
                if (MODE == 2) { \
                    B0 = dpp64<ROW_SHR(1)>(GG == 0 ? L : dpp64<ROW_SHL(GG ? 4 * GG : 1)>(L, L), B0); \
                    B1 = dpp64<ROW_SHR(1)>(dpp64<ROW_SHL(4 * GG + 1)>(L, L), B1); \
                    B2 = dpp64<ROW_SHR(1)>(dpp64<ROW_SHL(4 * GG + 2)>(L, L), B2); \
                    B3 = dpp64<ROW_SHR(1)>(dpp64<ROW_SHL(4 * GG + 3)>(L, L), B3); \
                } else { \
                    B0 = xp[0][-gi]; B1 = xp[1][-gi]; B2 = xp[2][-gi]; B3 = xp[3][-gi]; \
                } \
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A0, B0, acc, 0, 0, 0); \
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A1, B1, acc, 0, 0, 0); \
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A2, B2, acc, 0, 0, 0); \
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A3, B3, acc, 0, 0, 0); \
                hcur = hnext; hnext = hnn; }
            GROUP(0) GROUP(1) GROUP(2) GROUP(3)
#undef GROUP
            if (MODE == 2) L = Lnext;
        }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

template <int MODE>
void run(const char *name, int wps, double *out)
{
    const int ngroups = 64, reps = 5;
    int blocks = 256 * wps;
    size_t lds = 4000 * 8;
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), lds, 0, out, ngroups, 144);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), lds, 0, out, ngroups, 144);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double t = ms * 1e-3 / reps, nm = 64.0 * ngroups * 4;
    printf("%-44s waves/SIMD=%d : %8.2f us/launch  %6.1f cycles/MFMA/SIMD @2.4GHz  %6.2f TFLOP/s\n", name, wps, t * 1e6,
           t * 2.4e9 / (nm * wps), blocks * 4 * nm * 2048.0 / t / 1e12);
}

int main()
{
    double *out; hipMalloc(&out, 256 * 1024 * 8 * 8);
    for (int wps : {1, 2, 3, 4, 5}) {
        run<0>("both operands from LDS per MFMA", wps, out);
        run<1>("taps shifted with DPP, window from LDS", wps, out);
        run<2>("taps and window shifted with DPP", wps, out);
    }
    return 0;
}
