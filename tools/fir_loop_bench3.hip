// Third micro-benchmark: does the LDS *instruction form* matter?  Same 16-deep operand sets as
// fir_loop_bench2, but every operand is fetched by its own ds_read_b64 (2 LDS cycles per 512 B)
// written in inline asm so that hipcc cannot fuse pairs into ds_read2_b64 (8 LDS cycles per 1 KiB).
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 tools/fir_loop_bench3.hip -o /tmp/flb3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));

#define DSREAD(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(off))

template <int MODE>
__global__ __launch_bounds__(256) void k(double *out, int ngroups, int row)
{
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 9000; i += 256) lds[i] = 1.0 + i * 1e-6;
    __syncthreads();
    const int lane = threadIdx.x & 63, ij = lane & 15, kk = lane >> 4;
    // byte addresses in LDS
    unsigned hb0 = (unsigned)(size_t)(lds + 1 + kk + ij) ;
    hb0 = (1 + kk + ij) * 8;
    unsigned xb[4];
    for (int s = 0; s < 4; s++) xb[s] = (4300 + ((15 - kk - 4 * s) & 15) * row + 300 + ij) * 8;
    v4f64 acc = {0, 0, 0, 0};
    double ha[16], xa[16], hc[16], xc[16];
#define LOADSET(H, X, HB, XB0, XB1, XB2, XB3)                                                        \
    DSREAD(H[0], HB, 0);    DSREAD(H[1], HB, 32);   DSREAD(H[2], HB, 64);   DSREAD(H[3], HB, 96);    \
    DSREAD(H[4], HB, 128);  DSREAD(H[5], HB, 160);  DSREAD(H[6], HB, 192);  DSREAD(H[7], HB, 224);   \
    DSREAD(H[8], HB, 256);  DSREAD(H[9], HB, 288);  DSREAD(H[10], HB, 320); DSREAD(H[11], HB, 352);  \
    DSREAD(H[12], HB, 384); DSREAD(H[13], HB, 416); DSREAD(H[14], HB, 448); DSREAD(H[15], HB, 480);  \
    DSREAD(X[0], XB0, 24);  DSREAD(X[1], XB1, 24);  DSREAD(X[2], XB2, 24);  DSREAD(X[3], XB3, 24);   \
    DSREAD(X[4], XB0, 16);  DSREAD(X[5], XB1, 16);  DSREAD(X[6], XB2, 16);  DSREAD(X[7], XB3, 16);   \
    DSREAD(X[8], XB0, 8);   DSREAD(X[9], XB1, 8);   DSREAD(X[10], XB2, 8);  DSREAD(X[11], XB3, 8);   \
    DSREAD(X[12], XB0, 0);  DSREAD(X[13], XB1, 0);  DSREAD(X[14], XB2, 0);  DSREAD(X[15], XB3, 0);
#define WAITSET(H, X)                                                                                \
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(H[0]), "+v"(H[1]), "+v"(H[2]), "+v"(H[3]), "+v"(H[4]), "+v"(H[5]), "+v"(H[6]), "+v"(H[7]), \
                 "+v"(H[8]), "+v"(H[9]), "+v"(H[10]), "+v"(H[11]), "+v"(H[12]), "+v"(H[13]), "+v"(H[14]), "+v"(H[15]));                   \
    asm volatile("" : "+v"(X[0]), "+v"(X[1]), "+v"(X[2]), "+v"(X[3]), "+v"(X[4]), "+v"(X[5]), "+v"(X[6]), "+v"(X[7]),                     \
                 "+v"(X[8]), "+v"(X[9]), "+v"(X[10]), "+v"(X[11]), "+v"(X[12]), "+v"(X[13]), "+v"(X[14]), "+v"(X[15]));                   \
    __builtin_amdgcn_sched_barrier(0);
#define MFMASET(H, X)                                                                                \
    _Pragma("unroll") for (int u = 0; u < 16; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(H[u], X[u], acc, 0, 0, 0);

    unsigned hb = hb0;
    LOADSET(ha, xa, hb, xb[0], xb[1], xb[2], xb[3]);
    WAITSET(ha, xa);
    for (int rep = 0; rep < 32; rep++) {
        hb = hb0;
        unsigned x0 = xb[0], x1 = xb[1], x2 = xb[2], x3 = xb[3];
        for (int g = 0; g < ngroups; g += 8) {
            hb += 512; x0 -= 32; x1 -= 32; x2 -= 32; x3 -= 32;
            LOADSET(hc, xc, hb, x0, x1, x2, x3);
            __builtin_amdgcn_sched_barrier(0);
            MFMASET(ha, xa);
            __builtin_amdgcn_sched_barrier(0);
            WAITSET(hc, xc);
            hb += 512; x0 -= 32; x1 -= 32; x2 -= 32; x3 -= 32;
            LOADSET(ha, xa, hb, x0, x1, x2, x3);
            __builtin_amdgcn_sched_barrier(0);
            MFMASET(hc, xc);
            __builtin_amdgcn_sched_barrier(0);
            WAITSET(ha, xa);
        }
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
    for (int i = 0; i < 2; i++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, out, ngroups, 336);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, out, ngroups, 336);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double t = ms * 1e-3 / reps, nm = 32.0 * ngroups * 4;
    printf("%-44s waves/SIMD=%d : %8.2f us/launch  %6.1f cycles/MFMA/SIMD @2.4GHz  %6.2f TFLOP/s\n", name, wps, t * 1e6,
           t * 2.4e9 / (nm * wps), blocks * 4 * nm * 2048.0 / t / 1e12);
}

int main()
{
    double *out; hipMalloc(&out, 256 * 1024 * 8 * 8);
    for (int wps : {1, 2}) run<0>("16-deep sets, single ds_read_b64 per operand", wps, out);
    return 0;
}
