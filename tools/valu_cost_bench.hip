// Micro-benchmark: what the instructions of the cascade step cost one wave alone on its SIMD (gfx950), measured with
// s_memtime stamps around unrolled instruction sequences (no memory traffic).  Build and run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/valu_cost_bench.hip -o /tmp/valu_cost && /tmp/valu_cost
// Every sequence is N copies of a pattern in inline asm (the compiler cannot reorder or fold anything), looped 32 times
// from a warm instruction cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

enum { K_FMA_DEP, K_FMA_IND4, K_CVT32_DEP, K_CVT_PAIR_DEP, K_CVT64_IND, K_CVT32_IND, K_DPP_DEP, K_DPP_IND, K_MOV_DEP, K_MAD64_DEP,
       K_FMA_CVT_MIX, K_STEP_NAIVE, K_STEP_SKEW, K_STEP_INT, K_DP7_MOV1, K_DP7_DPP1, K_DP7_SP5_END, K_DP8_SP2, K_DP8_SP5_SPREAD, K_DP8_NOP, K_STEP_F32, K_COUNT };
const char *names[K_COUNT] = {
    "v_fma_f64, dependent chain", "v_fma_f64, 4 independent chains", "v_cvt_f32_f64 -> v_cvt_f64_f32 dependent (per pair)",
    "same, counted per instruction", "v_cvt_f64_f32 independent", "v_cvt_f32_f64 independent", "v_mov_b32 dpp row_shr:1 dependent",
    "v_mov_b32 dpp row_ror:1 independent x4", "v_mov_b32 dependent", "v_mad_i64_i32 dependent",
    "fma chain of 5 + cvt32 + cvt64 (7 instr, per group)", "cascade step, hand-off at the head of the chain (13 instr)",
    "cascade step, hand-off one step ahead (13 instr)", "int64 cascade step: 5 mad_i64 + shift + sat (per step)",
    "5 fma + cvt32 + cvt64, then ONE v_mov_b32 (8 instr)", "5 fma + cvt32 + cvt64, then ONE dpp (8 instr)",
    "5 fma + cvt32 + cvt64, then dpp dpp dpp mov cndmask together (12 instr)", "cvt64 + 5 fma + cvt32 + cvt64, then dpp + cndmask (10 instr)",
    "cvt64 + 5 fma + cvt32 + cvt64 with 5 SP ops spread between them (13 instr)", "5 fma + cvt32 + cvt64 + 5 x s_nop 1 spread (12 instr)",
    "the 13-instr step with every f64 op replaced by its f32 twin" };

template <int K>
__global__ __launch_bounds__(64) void bench(unsigned long long *out, double seed, int LOOPS)
{
    double a = seed + threadIdx.x * 1e-3, b = 0.999, c = 1e-3, d = 0.5, e = 0.25, f = 0.125;
    double a1 = a + 1, a2 = a + 2, a3 = a + 3;
    float x = (float)seed, y = 1.5f, z = 2.5f, w = 3.5f;
    unsigned u0 = threadIdx.x, u1 = 1, u2 = 2, u3 = 3;
    long long l = threadIdx.x, l2 = 5;
    int i0 = 12345, i1 = 777;
    unsigned long long t0 = 0, t1 = 0;
    for (int trip = 0; trip < LOOPS; trip++) {
        if (trip == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        if constexpr (K == K_FMA_DEP) asm volatile(REP64("v_fma_f64 %0, %0, %1, %2\n\t") : "+v"(a) : "v"(b), "v"(c));
        if constexpr (K == K_FMA_IND4)
            asm volatile(REP16("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5\n\t")
                         : "+v"(a), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));
        if constexpr (K == K_CVT32_DEP || K == K_CVT_PAIR_DEP) asm volatile(REP64("v_cvt_f32_f64 %1, %0\n\tv_cvt_f64_f32 %0, %1\n\t") : "+v"(a), "+v"(x));
        if constexpr (K == K_CVT64_IND)
            asm volatile(REP16("v_cvt_f64_f32 %0, %4\n\tv_cvt_f64_f32 %1, %5\n\tv_cvt_f64_f32 %2, %6\n\tv_cvt_f64_f32 %3, %7\n\t")
                         : "+v"(a), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(x), "v"(y), "v"(z), "v"(w));
        if constexpr (K == K_CVT32_IND)
            asm volatile(REP16("v_cvt_f32_f64 %0, %4\n\tv_cvt_f32_f64 %1, %5\n\tv_cvt_f32_f64 %2, %6\n\tv_cvt_f32_f64 %3, %7\n\t")
                         : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(a), "v"(a1), "v"(a2), "v"(a3));
        if constexpr (K == K_DPP_DEP) asm volatile(REP64("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t") : "+v"(u0));
        if constexpr (K == K_DPP_IND)
            asm volatile(REP16("v_mov_b32_dpp %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                               "v_mov_b32_dpp %2, %2 row_ror:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %3, %3 row_ror:1 row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));
        if constexpr (K == K_MOV_DEP) asm volatile(REP64("v_mov_b32 %1, %0\n\tv_mov_b32 %0, %1\n\t") : "+v"(u0), "+v"(u1));
        if constexpr (K == K_MAD64_DEP) asm volatile(REP64("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\t") : "+v"(l) : "v"(i0), "v"(i1) : "vcc");
        if constexpr (K == K_FMA_CVT_MIX)
            asm volatile(REP16("v_fma_f64 %0, %2, %3, %0\n\tv_fma_f64 %0, %2, %4, %0\n\tv_fma_f64 %0, %2, %5, %0\n\tv_fma_f64 %0, %2, %6, %0\n\tv_fma_f64 %0, %2, %7, %0\n\t"
                               "v_cvt_f32_f64 %1, %0\n\tv_cvt_f64_f32 %2, %1\n\t")
                         : "+v"(a), "+v"(x), "+v"(a1) : "v"(b), "v"(c), "v"(d), "v"(e), "v"(f));
        // the step as round 1 had it: [dpp(prev) -> cvt64 -> 5 fma -> cvt32] all in one dependent chain, + cvt64(y), 2 rotates, mov, cndmask
        if constexpr (K == K_STEP_NAIVE)
            asm volatile(REP16(
                "v_mov_b32_dpp %3, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                "v_cvt_f64_f32 %2, %3\n\t"
                "v_fma_f64 %0, %2, %7, %0\n\tv_fma_f64 %0, %4, %8, %0\n\tv_fma_f64 %0, %4, %9, %0\n\tv_fma_f64 %0, %4, %10, %0\n\tv_fma_f64 %0, %4, %11, %0\n\t"
                "v_mov_b32_dpp %5, %5 row_ror:15 row_mask:0xf bank_mask:0xf\n\t"
                "v_mov_b32_dpp %6, %6 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                "v_cvt_f32_f64 %1, %0\n\t"
                "v_cndmask_b32 %6, %6, %1, vcc\n\t"
                "v_mov_b32 %3, %5\n\t"
                "v_cvt_f64_f32 %4, %1\n\t")
                : "+v"(a), "+v"(x), "+v"(a1), "+v"(u0), "+v"(a2), "+v"(u1), "+v"(u2) : "v"(b), "v"(c), "v"(d), "v"(e), "v"(f) : "vcc");
        // the same instructions with the hand-off fetched one step ahead: the dpp and its cvt64 read LAST step's values (x_old), off the chain
        if constexpr (K == K_STEP_SKEW)
            asm volatile(REP16(
                "v_fma_f64 %0, %2, %7, %0\n\t"
                "v_mov_b32_dpp %3, %12 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                "v_fma_f64 %0, %4, %8, %0\n\t"
                "v_cvt_f64_f32 %2, %3\n\t"
                "v_fma_f64 %0, %4, %9, %0\n\t"
                "v_mov_b32_dpp %5, %5 row_ror:15 row_mask:0xf bank_mask:0xf\n\t"
                "v_fma_f64 %0, %4, %10, %0\n\t"
                "v_mov_b32_dpp %6, %6 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                "v_fma_f64 %0, %4, %11, %0\n\t"
                "v_mov_b32 %12, %1\n\t"
                "v_cvt_f32_f64 %1, %0\n\t"
                "v_cndmask_b32 %6, %6, %1, vcc\n\t"
                "v_cvt_f64_f32 %4, %1\n\t")
                : "+v"(a), "+v"(x), "+v"(a1), "+v"(u0), "+v"(a2), "+v"(u1), "+v"(u2) : "v"(b), "v"(c), "v"(d), "v"(e), "v"(f), "v"(y) : "vcc");
        if constexpr (K == K_STEP_INT)
            asm volatile(REP16(
                "v_mad_i64_i32 %0, vcc, %1, %2, %0\n\tv_mad_i64_i32 %0, vcc, %1, %3, %0\n\tv_mad_i64_i32 %0, vcc, %1, %2, %0\n\t"
                "v_mad_i64_i32 %0, vcc, %1, %3, %0\n\tv_mad_i64_i32 %0, vcc, %1, %2, %0\n\t"
                "v_ashrrev_i64 %4, 28, %0\n\t")
                : "+v"(l), "+v"(i0) : "v"(i1), "v"(i0), "v"(l2));
#define DP7 "v_fma_f64 %0, %2, %7, %0\n\tv_fma_f64 %0, %4, %8, %0\n\tv_fma_f64 %0, %4, %9, %0\n\tv_fma_f64 %0, %4, %10, %0\n\tv_fma_f64 %0, %4, %11, %0\n\tv_cvt_f32_f64 %1, %0\n\tv_cvt_f64_f32 %4, %1\n\t"
#define OPS : "+v"(a), "+v"(x), "+v"(a1), "+v"(u0), "+v"(a2), "+v"(u1), "+v"(u2) : "v"(b), "v"(c), "v"(d), "v"(e), "v"(f) : "vcc"
        if constexpr (K == K_DP7_MOV1) asm volatile(REP16(DP7 "v_mov_b32 %3, %1\n\t") OPS);
        if constexpr (K == K_DP7_DPP1) asm volatile(REP16(DP7 "v_mov_b32_dpp %3, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t") OPS);
        if constexpr (K == K_DP7_SP5_END)
            asm volatile(REP16(DP7 "v_mov_b32_dpp %3, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %5, %5 row_ror:15 row_mask:0xf bank_mask:0xf\n\t"
                               "v_mov_b32_dpp %6, %6 row_ror:1 row_mask:0xf bank_mask:0xf\n\tv_mov_b32 %3, %5\n\tv_cndmask_b32 %6, %6, %1, vcc\n\t") OPS);
        if constexpr (K == K_DP8_SP2)
            asm volatile(REP16("v_cvt_f64_f32 %2, %3\n\t" DP7 "v_mov_b32_dpp %3, %1 row_ror:1 row_mask:0xf bank_mask:0xf\n\tv_cndmask_b32 %3, %3, %5, vcc\n\t") OPS);
        if constexpr (K == K_DP8_SP5_SPREAD)
            asm volatile(REP16("v_cvt_f64_f32 %2, %3\n\t"
                               "v_fma_f64 %0, %2, %7, %0\n\tv_mov_b32_dpp %3, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                               "v_fma_f64 %0, %4, %8, %0\n\tv_mov_b32_dpp %5, %5 row_ror:15 row_mask:0xf bank_mask:0xf\n\t"
                               "v_fma_f64 %0, %4, %9, %0\n\tv_mov_b32_dpp %6, %6 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                               "v_fma_f64 %0, %4, %10, %0\n\tv_mov_b32 %3, %5\n\t"
                               "v_fma_f64 %0, %4, %11, %0\n\tv_cndmask_b32 %6, %6, %1, vcc\n\t"
                               "v_cvt_f32_f64 %1, %0\n\tv_cvt_f64_f32 %4, %1\n\t") OPS);
        if constexpr (K == K_DP8_NOP)
            asm volatile(REP16("v_fma_f64 %0, %2, %7, %0\n\ts_nop 1\n\tv_fma_f64 %0, %4, %8, %0\n\ts_nop 1\n\tv_fma_f64 %0, %4, %9, %0\n\ts_nop 1\n\t"
                               "v_fma_f64 %0, %4, %10, %0\n\ts_nop 1\n\tv_fma_f64 %0, %4, %11, %0\n\ts_nop 1\n\tv_cvt_f32_f64 %1, %0\n\tv_cvt_f64_f32 %4, %1\n\t") OPS);
        if constexpr (K == K_STEP_F32)
            asm volatile(REP16(
                "v_mov_b32_dpp %3, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                "v_mov_b32 %5, %3\n\t"
                "v_fma_f32 %1, %5, %7, %1\n\tv_fma_f32 %1, %5, %7, %1\n\tv_fma_f32 %1, %5, %7, %1\n\tv_fma_f32 %1, %5, %7, %1\n\tv_fma_f32 %1, %5, %7, %1\n\t"
                "v_mov_b32_dpp %5, %5 row_ror:15 row_mask:0xf bank_mask:0xf\n\t"
                "v_mov_b32_dpp %6, %6 row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                "v_mov_b32 %4, %1\n\t"
                "v_cndmask_b32 %6, %6, %1, vcc\n\t"
                "v_mov_b32 %3, %5\n\t"
                "v_mov_b32 %2, %1\n\t")
                : "+v"(a), "+v"(x), "+v"(u3), "+v"(u0), "+v"(w), "+v"(u1), "+v"(u2) : "v"(y) : "vcc");
    }
    asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    // keep every value alive
    double sink = a + a1 + a2 + a3 + x + y + z + w + (double)(u0 + u1 + u2 + u3) + (double)l + (double)l2 + i0 + i1;
    if (sink == 123.456) t1 = 0;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int K> void run(unsigned long long *d_out, int per_rep, int reps)
{
    const int nblk = 256;
    hipLaunchKernelGGL(bench<K>, dim3(nblk), dim3(64), 0, 0, d_out, 1.25, 33);
    hipDeviceSynchronize();
    // wall clock: the same pattern 20001 times, against 1 time (launch overhead cancels)
    hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    hipLaunchKernelGGL(bench<K>, dim3(nblk), dim3(64), 0, 0, d_out, 1.25, 20001);
    hipEventRecord(e0);
    hipLaunchKernelGGL(bench<K>, dim3(nblk), dim3(64), 0, 0, d_out, 1.25, 20001);
    hipEventRecord(e1);
    hipLaunchKernelGGL(bench<K>, dim3(nblk), dim3(64), 0, 0, d_out, 1.25, 1);
    hipEventRecord(e2); hipEventSynchronize(e2);
    float ms_long, ms_short; hipEventElapsedTime(&ms_long, e0, e1); hipEventElapsedTime(&ms_short, e1, e2);
    const double ns_group = (ms_long - ms_short) * 1e6 / 20000.0 / reps;
    hipLaunchKernelGGL(bench<K>, dim3(nblk), dim3(64), 0, 0, d_out, 1.25, 33);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nblk);
    hipMemcpy(h.data(), d_out, nblk * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double cyc = (double)h[nblk / 2] / 32.0;        // per trip
    printf("%-66s %6.2f ticks per instruction  %7.2f per group | wall %7.2f ns per group = %6.1f cycles @2.4GHz (%.2f ns per tick)\n", names[K],
           cyc / (double)(per_rep * reps), cyc / reps, ns_group, ns_group * 2.4, ns_group / (cyc / reps));
}

int main()
{
    unsigned long long *d_out; hipMalloc(&d_out, 256 * 8);
    printf("one wave per SIMD (64-thread workgroups, 256 of them), median over workgroups, s_memtime cycles (shader clock)\n");
    run<K_FMA_DEP>(d_out, 1, 64); run<K_FMA_IND4>(d_out, 4, 16); run<K_CVT32_DEP>(d_out, 1, 64); run<K_CVT_PAIR_DEP>(d_out, 2, 64);
    run<K_CVT64_IND>(d_out, 4, 16); run<K_CVT32_IND>(d_out, 4, 16); run<K_DPP_DEP>(d_out, 1, 64); run<K_DPP_IND>(d_out, 4, 16);
    run<K_MOV_DEP>(d_out, 2, 64); run<K_MAD64_DEP>(d_out, 1, 64); run<K_FMA_CVT_MIX>(d_out, 7, 16); run<K_STEP_NAIVE>(d_out, 13, 16);
    run<K_STEP_SKEW>(d_out, 13, 16); run<K_STEP_INT>(d_out, 6, 16);
    run<K_DP7_MOV1>(d_out, 8, 16); run<K_DP7_DPP1>(d_out, 8, 16); run<K_DP7_SP5_END>(d_out, 12, 16); run<K_DP8_SP2>(d_out, 10, 16);
    run<K_DP8_SP5_SPREAD>(d_out, 13, 16); run<K_DP8_NOP>(d_out, 12, 16); run<K_STEP_F32>(d_out, 13, 16);
    return 0;
}
