// What does a step of the cascade's fill / drain cost, element by element?  (round 5)  One wave alone on its SIMD, the steady step of
// biquad_row on fixed registers, 16 copies per trip, 32 timed trips; variants add what the block's ends add.
//   hipcc -O2 --offload-arch=gfx950 tools/edge_step_lab.hip -o /tmp/edge_lab && /tmp/edge_lab
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define INIT "v_mov_b32 v10, 0\n\tv_mov_b32 v11, 0x3ff00000\n\tv_mov_b32 v12, 0\n\tv_mov_b32 v13, 0x3fe00000\n\tv_mov_b32 v14, 0\n\tv_mov_b32 v15, 0x3fe00000\n\t" \
             "v_mov_b32 v16, 0\n\tv_mov_b32 v17, 0x3fd00000\n\tv_mov_b32 v18, 0\n\tv_mov_b32 v19, 0x3fd00000\n\tv_mov_b32 v20, 0\n\tv_mov_b32 v21, 0x3fd00000\n\t" \
             "v_mov_b32 v22, 0\n\tv_mov_b32 v23, 0x3fd00000\n\tv_mov_b32 v24, 0\n\tv_mov_b32 v25, 0x3fd00000\n\tv_mov_b32 v26, 0\n\tv_mov_b32 v27, 0x3fd00000\n\t" \
             "v_mov_b32 v30, 0x3f000000\n\tv_mov_b32 v32, 0x3e000000\n\tv_mov_b32 v40, 0\n\tv_mov_b32 v41, 0x3fb00000\n\tv_mov_b32 v42, 0\n\tv_mov_b32 v43, 0x3fb00000\n\t" \
             "s_mov_b32 s20, 0x00010001\n\ts_mov_b32 s21, 0x00010001\n\ts_mov_b64 s[22:23], -1\n\ts_mov_b64 s[24:25], 0\n\ts_mov_b32 s26, 0x00f000f0\n\ts_mov_b32 s27, 0x00f000f0\n\ts_mov_b32 s30, 0x7fffffff\n\t"
#define CLOB "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v30","v31","v32","v33","v34","v40","v41","v42","v43","v44","v45","v46","v47","vcc","scc","s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","s30"
// the steady step: t = handoff; an = a2 + x1 c1 + x2 c2 + y1 c3 + y2 c4; P = widen(t); o = (float)an; a2 = an + P c0; y = widen(o); bcast
#define STEP "v_cndmask_b32_dpp v31, v30, v32, vcc row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
             "v_fma_f64 v[12:13], v[14:15], v[40:41], v[10:11]\n\tv_fma_f64 v[12:13], v[16:17], v[42:43], v[12:13]\n\tv_fma_f64 v[12:13], v[18:19], v[40:41], v[12:13]\n\tv_fma_f64 v[12:13], v[20:21], v[42:43], v[12:13]\n\t" \
             "v_cvt_f64_f32 v[16:17], v31\n\tv_cvt_f32_f64 v30, v[12:13]\n\tv_fma_f64 v[10:11], v[16:17], v[40:41], v[12:13]\n\tv_cvt_f64_f32 v[20:21], v30\n\t" \
             "v_mov_b32_dpp v33, v30 row_newbcast:15 row_mask:0xf bank_mask:0x1\n\t"
#define VCC "s_mov_b64 vcc, s[20:21]\n\t"
#define HOOK_SKIP "s_mov_b64 s[28:29], exec\n\ts_mov_b64 exec, s[24:25]\n\ts_cbranch_execz 1f\n\tv_mov_b64 v[22:23], v[14:15]\n\t1:\n\ts_mov_b64 exec, s[28:29]\n\t"
#define HOOK_NOBR "s_mov_b64 s[28:29], exec\n\ts_mov_b64 exec, s[26:27]\n\tv_mov_b64 v[22:23], v[14:15]\n\tv_mov_b64 v[24:25], v[16:17]\n\tv_mov_b64 v[26:27], v[18:19]\n\tv_mov_b64 v[44:45], v[20:21]\n\tv_mov_b64 v[46:47], v[12:13]\n\tv_mov_b32 v34, v30\n\ts_mov_b64 exec, s[28:29]\n\t"
#define EXEC2 "s_mov_b64 s[28:29], exec\n\ts_mov_b64 exec, s[22:23]\n\ts_mov_b64 exec, s[28:29]\n\t"
// the masked step as shipped (three EXEC writes, three moves, s_nop)
#define MASKED "s_mov_b64 s[28:29], exec\n\ts_mov_b64 vcc, s[20:21]\n\tv_cndmask_b32_dpp v31, v30, v32, vcc row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_mov_b64 exec, s[22:23]\n\t" \
             "v_fma_f64 v[10:11], v[14:15], v[40:41], v[10:11]\n\tv_fma_f64 v[10:11], v[16:17], v[42:43], v[10:11]\n\tv_fma_f64 v[10:11], v[18:19], v[40:41], v[10:11]\n\tv_fma_f64 v[10:11], v[20:21], v[42:43], v[10:11]\n\t" \
             "v_cvt_f32_f64 v30, v[10:11]\n\tv_mov_b64 v[16:17], v[14:15]\n\tv_mov_b64 v[14:15], v[22:23]\n\tv_mov_b64 v[20:21], v[18:19]\n\tv_cvt_f64_f32 v[18:19], v30\n\ts_mov_b64 exec, s[22:23]\n\t" \
             "v_cvt_f64_f32 v[22:23], v31\n\ts_nop 0\n\tv_fma_f64 v[10:11], v[22:23], v[40:41], v[10:11]\n\ts_mov_b64 exec, s[28:29]\n\t" \
             "v_mov_b32_dpp v33, v30 row_newbcast:15 row_mask:0xf bank_mask:0x1\n\t"
#define VCMP "v_cmp_gt_u32_e64 s[22:23], s30, v32\n\t"     /* (the shipped step's mask: a VALU compare into an SGPR pair, all lanes true here) */
template <int K>
__global__ __launch_bounds__(64) void lab(unsigned long long *out, double *sink)
{
    unsigned long long t0 = 0, t1 = 0; unsigned r32 = 0;
    for (int trip = 0; trip < 33; trip++) {
        if (trip == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        if constexpr (K == 0) asm volatile(INIT VCC REP16(STEP) "v_mov_b32 %0, v10" : "=v"(r32) :: CLOB);
        if constexpr (K == 1) asm volatile(INIT REP16(VCC STEP) "v_mov_b32 %0, v10" : "=v"(r32) :: CLOB);
        if constexpr (K == 2) asm volatile(INIT REP16(VCC STEP EXEC2) "v_mov_b32 %0, v10" : "=v"(r32) :: CLOB);
        if constexpr (K == 3) asm volatile(INIT REP16(VCC STEP HOOK_SKIP) "v_mov_b32 %0, v10" : "=v"(r32) :: CLOB);
        if constexpr (K == 4) asm volatile(INIT REP16(VCC STEP HOOK_NOBR) "v_mov_b32 %0, v10" : "=v"(r32) :: CLOB);
        if constexpr (K == 5) asm volatile(INIT REP16(MASKED) "v_mov_b32 %0, v10" : "=v"(r32) :: CLOB);
        if constexpr (K == 6) asm volatile(INIT REP16(VCMP MASKED) "v_mov_b32 %0, v10" : "=v"(r32) :: CLOB);
        if constexpr (K == 7) asm volatile(INIT REP16(VCMP VCMP MASKED) "v_mov_b32 %0, v10" : "=v"(r32) :: CLOB);
    }
    asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + threadIdx.x] = r32;
}
template <int K> static void run(const char *name)
{
    unsigned long long *d; double *sink; CHECK(hipMalloc(&d, 1024 * 8)); CHECK(hipMalloc(&sink, 1024 * 64 * 8));
    for (int i = 0; i < 2; i++) { hipLaunchKernelGGL(lab<K>, dim3(1024), dim3(64), 0, 0, d, sink); CHECK(hipDeviceSynchronize()); }
    std::vector<unsigned long long> h(1024); CHECK(hipMemcpy(h.data(), d, 1024 * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    printf("%-78s %7.1f cycles per step (median of 1024 waves, one per SIMD; min %.1f)\n", name, h[512] / (32.0 * 16), h[0] / (32.0 * 16));
    CHECK(hipFree(d)); CHECK(hipFree(sink));
}
int main()
{
    run<0>("steady step, 16 in a row (10 instructions)");
    run<1>("+ s_mov vcc per step");
    run<2>("+ s_mov vcc + save / write / restore EXEC (no branch, nothing under it)");
    run<3>("+ s_mov vcc + hook skipped by s_cbranch_execz (EXEC = 0)");
    run<4>("+ s_mov vcc + hook taken: six moves under an EXEC mask of four lanes");
    run<5>("the masked step as shipped (3 EXEC writes, 3 moves, s_nop), masks from SGPRs");
    run<6>("the masked step + one v_cmp into the mask's SGPR pair in front (as shipped)");
    run<7>("the masked step + two v_cmp");
    return 0;
}
