// cascade_lab: candidate forms of the section-pipelined biquad step (DSP_FORMAT 6, 16 lanes per channel), standalone, checked
// against a CPU loop and timed with HIP events.  Not part of the product library: the form that wins goes into
// avdsp_kernels.hip (biquad_row).  Build and run on the GPU box:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/cascade_lab.hip -o /tmp/cascade_lab && /tmp/cascade_lab
//
// Variants (template VAR):
//   0  hand-off = v_mov_b32_dpp row_shr:1 + v_cvt_f64_f32 (9 VALU per step), operand fetched one step ahead (skew 2)
//   1  hand-off = v_cvt_f64_f32 WITH the dpp modifier, hand-encoded (the assembler refuses DP-ALU dpp other than
//      row_newbcast; does the hardware?) (8 VALU per step), skew 2
//   2  as 0 with the hand-off at the head of the step that uses it (skew 1: half the fill / drain steps)
//   3  as 1, skew 1
// Sample IO of all variants: a batch of 16 frames per row goes global -> LDS -> 16 registers of every lane (lane 0 of a row uses
// them: the dpp hand-off leaves lane 0's register alone), results stay in 16 registers of the last-section lane and leave
// through LDS as one store per batch.  No rotating batch registers, no select.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <type_traits>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args {
    const float *coef;      // [C][16][5]
    int *state;             // [C][16][6]: acc lo, acc hi, x1, x2, y1, y2
    const float *in;        // [B][C]
    float *out;             // [B][C]
    int C, B, nsec;
    unsigned long long *stamps;     // [wave][4]: s_memtime / s_memrealtime at the wave's start and end (VAR & 8)
    float *out2;                    // [C][B + 64]: where the steady batches put their results in the "ring-like" output modes (OM >= 2)
};

template <int CTRL>
__device__ __forceinline__ unsigned dpp_mov(unsigned old, unsigned src)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xF, 0xF, false);
}
constexpr int kRowShr1 = 0x111;

// v_cvt_f64_f32_dpp d, s row_shr:1 row_mask:0xf bank_mask:0xf  -- hand-encoded: VOP1 (0x7e << 24 | vdst << 17 | op 0x10 << 9 | 0xfa),
// then src0 | dpp_ctrl 0x111 << 8 | row_mask/bank_mask 0xff << 24.  Fixed registers: d = v[2:3], s = v4.
__device__ __forceinline__ double cvt_dpp_shr1(double old, unsigned src)
{
    double d;
    asm volatile(".long 0x7e0420fa\n\t.long 0xff011104" : "={v[2:3]}"(d) : "{v4}"(src), "0"(old));
    return d;
}

__device__ __forceinline__ void flush_mode() { __builtin_amdgcn_s_setreg(1 | (4 << 6) | (1 << 11), 0); }

// Step conventions: a lane keeps {acc, P, dx1, dx2, dy1, dy2, hy}:
//   acc   the accumulator, with the first product of the lane's NEXT compute (P * c0) already added if it computes next step
//   P     the operand of that product (becomes x1 once used), dx1 / dx2 the delayed inputs, dy1 / dy2 the delayed outputs, as doubles
//   hy    the lane's latest result as float bits (= y1)
// step u:  t = hand-off from lane-1 (its hy; a section-0 lane: the staged input), dxn = widen(t)
//          lanes computing index u-1-2 sec:  acc += dx1 c1 + dx2 c2 + dy1 c3 + dy2 c4 (in that order, one fma each);
//                                            hy = (float)acc; dx2 = dx1; dx1 = P; dy2 = dy1; dy1 = widen(hy)
//          lanes computing next step:        acc += dxn c0; P = dxn
// Rows are right-aligned: a chain's last section sits in lane 15 of its row, whatever the section count, so that its result
// reaches the lane that stores it with one dpp row_newbcast:15 per step (four registers, one bank of four lanes each) and three
// selects per batch; section 0 sits in lane 16 - nsec and takes the staged input instead of the hand-off (a lane mask in vcc).
// VAR bit 0: plain instruction order; bit 3: clock stamps.  OM: 0 = results through LDS (4 x ds_write_b128), 6 = row_newbcast.
template <int VAR, int OM = 6>
__global__ __launch_bounds__(256) void bq_lab(const Args a)
{
    flush_mode();
    unsigned long long t_c0 = 0, t_r0 = 0;
    if constexpr (VAR & 8) { t_c0 = __builtin_amdgcn_s_memtime(); t_r0 = __builtin_amdgcn_s_memrealtime(); }
    __shared__ __attribute__((aligned(16))) float lin[4][2][4][16];
    __shared__ __attribute__((aligned(16))) float lout[4][64 * 20];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, row = lane >> 4, rp = lane & 15;
    const int ch = (blockIdx.x * 4 + wv) * 4 + row;
    const int B = a.B, C = a.C, L = a.nsec - 1;
    const int s = rp - (15 - L);                    // section of this lane (negative: none)
    const bool lane_on = ch < C && s >= 0;
    const int chc = ch < C ? ch : C - 1;
    const unsigned long long firstmask = 0x0001000100010001ull << (15 - L);        // the section-0 lanes

    double cd[5] = {0, 0, 0, 0, 0}, acc = 0, dx1 = 0, dx2 = 0, dy1 = 0, dy2 = 0, P = 0;
    unsigned hy = 0;
    if (lane_on) {
        const float *co = a.coef + ((size_t)chc * 16 + s) * 5;
        const int *st = a.state + ((size_t)chc * 16 + s) * 6;
        for (int k = 0; k < 5; k++) cd[k] = (double)co[k];
        acc = __longlong_as_double(((long long)st[1] << 32) | (unsigned)st[0]);
        dx1 = (double)__int_as_float(st[2]); dx2 = (double)__int_as_float(st[3]);
        hy = (unsigned)st[4]; dy1 = (double)__int_as_float(st[4]); dy2 = (double)__int_as_float(st[5]);
    }
    // input: lane (row, i) fetches frame 16 b + i of its row's channel
    const float *inp = a.in + chc;
    auto fetch = [&](int b) -> float { int n = 16 * b + rp; n = n < B ? n : B - 1; return inp[(size_t)n * C]; };
    float rawq[3] = {fetch(0), fetch(1), fetch(2)};
    float *mylin = &lin[wv][0][row][0];
    typedef float f4 __attribute__((ext_vector_type(4)));
    auto stage = [&](int buf, float v) { mylin[buf * 64 + rp] = v; };
    auto take = [&](int buf, unsigned (&x)[16]) {
        const f4 *p = reinterpret_cast<const f4 *>(mylin + buf * 64);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f4 v = p[q];
#pragma unroll
            for (int j = 0; j < 4; j++) x[4 * q + j] = __float_as_uint(v[j]);
        }
    };
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0};
    if constexpr (VAR & 8) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); ph[0] = __builtin_amdgcn_s_memtime(); }
    unsigned xa[16], xb[16];
    stage(0, rawq[0]);
    rawq[0] = fetch(3);
    take(0, xa);
    if constexpr (VAR & 8) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); ph[1] = __builtin_amdgcn_s_memtime(); }

    const int U = B + 2 * L + 1;                    // steps u = 0 .. U-1; section s computes index n = u - 1 - 2 s
    const int nb = (U + 15) / 16;
    float *mylout = &lout[wv][0];
    const float *outp_src = mylout + (row * 16 + 15) * 20 + rp;
    float *outp = a.out + chc;
    const float *in_run = nullptr;
    float *out_run = nullptr;
    float wprev = 0.f;
    const bool q1 = (rp & 1) != 0, q2 = (rp & 2) != 0;

    auto batch = [&](int b, unsigned (&x)[16], unsigned (&xn)[16], int slot, auto only_steady) __attribute__((always_inline)) {
        constexpr bool ST = decltype(only_steady)::value;
        // stage the next batch's inputs and request the one after the prefetch window
        stage((b + 1) & 1, rawq[(slot + 1) % 3]);
        if constexpr (ST) { rawq[(slot + 1) % 3] = *in_run; in_run += (size_t)16 * C; }
        else rawq[(slot + 1) % 3] = fetch(b + 4);
        take((b + 1) & 1, xn);
        unsigned yo[16];
        unsigned d0 = 0, d1 = 0, d2 = 0, d3 = 0;     // row_newbcast targets: dj lane 4 q + i = result of step 4 q + j
        const int u0 = 16 * b;
        const bool steady = ST || (u0 >= 2 * L + 1 && u0 + 15 <= B - 1);      // every lane computes in all 16 steps AND in the step after
        if (steady) {
#define DPP " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
#define BCAST(D, O, BM) "v_mov_b32_dpp %[" D "], %[" O "] row_newbcast:15 row_mask:0xf bank_mask:" BM "\n\t"
            // the hand-off, the four remaining products, the new operand, the rounding, the first product of the next step, the widening
            // of the result, [its broadcast]; no instruction directly behind one whose result it needs except fma -> fma
#define STEPP(X1, X2, YA, YB, XK, HY, O, D, BM) \
            "v_cndmask_b32_dpp %[t" XK "], %[" HY "], %[" XK "], vcc" DPP \
            "v_fma_f64 %[an], %[" X1 "], %[c1], %[an]\n\t" \
            "v_fma_f64 %[an], %[" X2 "], %[c2], %[an]\n\t" \
            "v_fma_f64 %[an], %[" YA "], %[c3], %[an]\n\t" \
            "v_fma_f64 %[an], %[" YB "], %[c4], %[an]\n\t" \
            "v_cvt_f64_f32 %[" X2 "], %[t" XK "]\n\t" \
            "v_cvt_f32_f64 %[" O "], %[an]\n\t" \
            "v_fma_f64 %[an], %[" X2 "], %[c0], %[an]\n\t" \
            "v_cvt_f64_f32 %[" YB "], %[" O "]\n\t" \
            BCAST(D, O, BM)
#define STEPN(X1, X2, YA, YB, XK, HY, O, D, BM) \
            "v_cndmask_b32_dpp %[t" XK "], %[" HY "], %[" XK "], vcc" DPP \
            "v_fma_f64 %[an], %[" X1 "], %[c1], %[an]\n\t" \
            "v_fma_f64 %[an], %[" X2 "], %[c2], %[an]\n\t" \
            "v_fma_f64 %[an], %[" YA "], %[c3], %[an]\n\t" \
            "v_fma_f64 %[an], %[" YB "], %[c4], %[an]\n\t" \
            "v_cvt_f64_f32 %[" X2 "], %[t" XK "]\n\t" \
            "v_cvt_f32_f64 %[" O "], %[an]\n\t" \
            "v_fma_f64 %[an], %[" X2 "], %[c0], %[an]\n\t" \
            "v_cvt_f64_f32 %[" YB "], %[" O "]\n\t"
#define BLOCK4(S, BM) \
                asm volatile( \
                    "s_mov_b64 vcc, %[m0]\n\t" \
                    S("xb", "xc", "ya", "yb", "x0", "hy", "o0", "d0", BM) \
                    S("xa", "xb", "yb", "ya", "x1", "o0", "o1", "d1", BM) \
                    S("xc", "xa", "ya", "yb", "x2", "o1", "o2", "d2", BM) \
                    S("xb", "xc", "yb", "ya", "x3", "o2", "o3", "d3", BM) \
                    : [an] "+v"(acc), [xa] "+v"(P), [xb] "+v"(dx1), [xc] "+v"(dx2), [ya] "+v"(dy1), [yb] "+v"(dy2), \
                      [tx0] "=&v"(t0), [tx1] "=&v"(t1), [tx2] "=&v"(t2), [tx3] "=&v"(t3), \
                      [o0] "=&v"(yo[k]), [o1] "=&v"(yo[k + 1]), [o2] "=&v"(yo[k + 2]), [o3] "=&v"(yo[k + 3]), \
                      [d0] "+v"(d0), [d1] "+v"(d1), [d2] "+v"(d2), [d3] "+v"(d3) \
                    : [hy] "v"(hy), [c0] "v"(cd[0]), [c1] "v"(cd[1]), [c2] "v"(cd[2]), [c3] "v"(cd[3]), [c4] "v"(cd[4]), \
                      [x0] "v"(x[k]), [x1] "v"(x[k + 1]), [x2] "v"(x[k + 2]), [x3] "v"(x[k + 3]), [m0] "s"(firstmask) \
                    : "vcc")
#pragma unroll
            for (int k = 0; k < 16; k += 4) {
                unsigned t0, t1, t2, t3;
                // roles: a step turns the (P, x1, x2) registers (r0, r1, r2) into (r2, r0, r1) and swaps (y1, y2)
                if constexpr (OM == 6) {
                    if (k == 0) BLOCK4(STEPP, "0x1"); else if (k == 4) BLOCK4(STEPP, "0x2"); else if (k == 8) BLOCK4(STEPP, "0x4"); else BLOCK4(STEPP, "0x8");
                } else BLOCK4(STEPN, "0");
                { const double t = P; P = dx2; dx2 = dx1; dx1 = t; }
                hy = yo[k + 3];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const int u = u0 + k;
                const unsigned hand = dpp_mov<kRowShr1>(0u, hy);
                const double dxn = (double)__uint_as_float(s == 0 ? x[k] : hand);
                const int n = u - 1 - 2 * s;
                if (lane_on && n >= 0 && n < B) {
                    acc = __builtin_fma(dx1, cd[1], acc);
                    acc = __builtin_fma(dx2, cd[2], acc);
                    acc = __builtin_fma(dy1, cd[3], acc);
                    acc = __builtin_fma(dy2, cd[4], acc);
                    hy = __float_as_uint((float)acc);
                    dx2 = dx1; dx1 = P; dy2 = dy1; dy1 = (double)__uint_as_float(hy);
                }
                if (lane_on && n + 1 >= 0 && n + 1 < B) { acc = __builtin_fma(dxn, cd[0], acc); P = dxn; }
                yo[k] = hy;
            }
        }
        float w;
        if (OM == 6 && steady) {
            // lane 4 q + i of a row: the result of step 4 q + i is in d_i
            const unsigned lo = q1 ? d1 : d0, hi = q1 ? d3 : d2;
            w = __uint_as_float(q2 ? hi : lo);
        } else {
            // results leave through LDS: every lane parks its 16, lane (row, i) picks up frame i of the row's last section
            f4 *q = reinterpret_cast<f4 *>(mylout + lane * 20);
#pragma unroll
            for (int j = 0; j < 4; j++) q[j] = f4{__uint_as_float(yo[4 * j]), __uint_as_float(yo[4 * j + 1]), __uint_as_float(yo[4 * j + 2]), __uint_as_float(yo[4 * j + 3])};
            w = *outp_src;
        }
        if constexpr (ST && OM == 6) { if (ch < C) *out_run = w; out_run += (size_t)16 * C; }
        else {
            // (a value picked up from LDS is stored a batch later: its latency stays off the path)
            const int back = OM == 6 ? 0 : 16;
            const float v = OM == 6 ? w : wprev;
            if constexpr (ST) { if (ch < C) *out_run = v; out_run += (size_t)16 * C; }
            else {
                const int n = u0 - back + rp - (1 + 2 * L);
                if (ch < C && n >= 0 && n < B) outp[(size_t)n * C] = v;
            }
            wprev = w;
        }
    };
    auto six = [&](int b, auto only_steady) __attribute__((always_inline)) {
        constexpr bool st = decltype(only_steady)::value;
        batch(b, xa, xb, 0, only_steady); if (!st && b + 1 >= nb) return;
        batch(b + 1, xb, xa, 1, only_steady); if (!st && b + 2 >= nb) return;
        batch(b + 2, xa, xb, 2, only_steady); if (!st && b + 3 >= nb) return;
        batch(b + 3, xb, xa, 0, only_steady); if (!st && b + 4 >= nb) return;
        batch(b + 4, xa, xb, 1, only_steady); if (!st && b + 5 >= nb) return;
        batch(b + 5, xb, xa, 2, only_steady);
    };
    const int first_steady = (2 * L + 1 + 15) / 16;             // batches b with 16 b >= 2 L + 1 ...
    const int end_steady = B >= 16 ? (B - 16) / 16 + 1 : 0;     // ... and 16 b + 15 <= B - 1: b < end_steady
    const int bs = (first_steady + 5) / 6 * 6;
    int b = 0;
    for (; b < nb && b < bs; b += 6) six(b, std::false_type{});
    if constexpr (VAR & 8) ph[2] = __builtin_amdgcn_s_memtime();
    if (b + 6 <= end_steady - 4) {                  /* (- 4: the running prefetch pointer stays inside the block) */
        in_run = inp + (size_t)(16 * (b + 4) + rp) * C;
        out_run = outp + (size_t)(16 * (b - (OM == 6 ? 0 : 1)) + rp - (1 + 2 * L)) * C;
        for (; b + 6 <= end_steady - 4; b += 6) six(b, std::true_type{});
    }
    if constexpr (VAR & 8) ph[3] = __builtin_amdgcn_s_memtime();
    for (; b < nb; b += 6) six(b, std::false_type{});
    if constexpr (VAR & 8) ph[4] = __builtin_amdgcn_s_memtime();
    if (OM != 6) {
        const int n = 16 * (nb - 1) + rp - (1 + 2 * L);
        if (ch < C && n >= 0 && n < B) outp[(size_t)n * C] = wprev;
    }
    if (lane_on) {
        int *st = a.state + ((size_t)chc * 16 + s) * 6;
        const unsigned long long bits = (unsigned long long)__double_as_longlong(acc);
        st[0] = (int)(unsigned)bits; st[1] = (int)(unsigned)(bits >> 32);
        st[2] = __float_as_int((float)dx1); st[3] = __float_as_int((float)dx2);
        st[4] = (int)hy; st[5] = __float_as_int((float)dy2);
    }
    if constexpr (VAR & 8) {
        const unsigned long long t_c1 = __builtin_amdgcn_s_memtime(), t_r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            unsigned long long *q = a.stamps + (size_t)(blockIdx.x * 4 + wv) * 4;
            q[0] = t_c0; q[1] = t_r0; q[2] = t_c1; q[3] = t_r1;
            unsigned long long *pq = a.stamps + (size_t)gridDim.x * 16 + (size_t)(blockIdx.x * 4 + wv) * 8;
            for (int i = 0; i < 5; i++) pq[i] = ph[i] - t_c0;
        }
    }
}

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
// issue cost of single instructions from one wave alone on its SIMD (s_memtime around 64 copies, 32 warm trips)
template <int K>
__global__ __launch_bounds__(64) void micro(unsigned long long *out, double *sink)
{
    unsigned long long t0 = 0, t1 = 0;
    double acc = threadIdx.x * 0.5, r = 0; unsigned r32 = 0; long long lacc = threadIdx.x;
    for (int trip = 0; trip < 33; trip++) {
        if (trip == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        if constexpr (K == 0) asm volatile("v_mov_b32 v4, %1\n\t" REP64("v_cvt_f64_f32 v[2:3], v4\n\t") "v_mov_b32 %0, v2" : "=v"(r32) : "v"((float)acc) : "v2", "v3", "v4");
        if constexpr (K == 1) asm volatile("v_mov_b32 v4, %1\n\t" REP64(".long 0x7e0420fa\n\t.long 0xff011104\n\t") "v_mov_b32 %0, v2" : "=v"(r32) : "v"((float)acc) : "v2", "v3", "v4");
        if constexpr (K == 2) asm volatile("v_mov_b32 v4, %1\n\t" REP64("v_mov_b32_dpp v5, v4 row_shr:1 row_mask:0xf bank_mask:0xf\n\t") "v_mov_b32 %0, v5" : "=v"(r32) : "v"((float)acc) : "v4", "v5");
        if constexpr (K == 3) asm volatile(REP64("v_fma_f64 %0, %0, %1, %2\n\t") : "+v"(acc) : "v"(0.999), "v"(1e-3));
        if constexpr (K == 4) asm volatile(REP64("v_mad_i64_i32 %0, vcc, %1, %2, %0\n\t") : "+v"(lacc) : "v"(12345), "v"(777) : "vcc");
        if constexpr (K == 5) asm volatile("v_mov_b32 v4, %1\n\t" REP64("v_cvt_f32_f64 v5, v[2:3]\n\tv_cvt_f64_f32 v[2:3], v5\n\t") "v_mov_b32 %0, v2" : "=v"(r32) : "v"((float)acc) : "v2", "v3", "v4", "v5");
        if constexpr (K == 6) asm volatile(REP64("v_alignbit_b32 %0, %1, %0, 28\n\t") : "+v"(r32) : "v"(777));
        if constexpr (K == 7) asm volatile(REP64("s_nop 0\n\t"));
        if constexpr (K == 8) asm volatile(REP64("s_mov_b64 s[20:21], -1\n\t") ::: "s20", "s21");
        if constexpr (K == 9) asm volatile(REP64("v_mov_b64 %0, %1\n\t") : "+v"(r) : "v"(acc));
        // register banks: the three 64-bit operands of a v_fma_f64 in the same / different halves of the four VGPR banks (register number mod 4)
#define FMA_INIT "v_mov_b32 v10, 0\n\tv_mov_b32 v11, 0x3ff00000\n\tv_mov_b32 v12, 0\n\tv_mov_b32 v13, 0x3fe00000\n\tv_mov_b32 v14, 0\n\tv_mov_b32 v15, 0x3fe00000\n\t" \
                 "v_mov_b32 v16, 0\n\tv_mov_b32 v17, 0x3fd00000\n\tv_mov_b32 v18, 0\n\tv_mov_b32 v19, 0x3fd00000\n\tv_mov_b32 v20, 0\n\tv_mov_b32 v21, 0x3fd00000\n\t"
#define FMA_CLOB "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21"
        if constexpr (K == 10) asm volatile(FMA_INIT REP64("v_fma_f64 v[10:11], v[14:15], v[18:19], v[10:11]\n\t") "v_mov_b32 %0, v10" : "=v"(r32) :: FMA_CLOB);   // all in banks 2,3
        if constexpr (K == 11) asm volatile(FMA_INIT REP64("v_fma_f64 v[10:11], v[12:13], v[16:17], v[10:11]\n\t") "v_mov_b32 %0, v10" : "=v"(r32) :: FMA_CLOB);   // acc 2,3; sources both 0,1
        if constexpr (K == 12) asm volatile(FMA_INIT REP64("v_fma_f64 v[10:11], v[12:13], v[18:19], v[10:11]\n\t") "v_mov_b32 %0, v10" : "=v"(r32) :: FMA_CLOB);   // acc 2,3; one source 0,1, one 2,3
        if constexpr (K == 13) asm volatile(FMA_INIT REP64("v_fma_f64 v[10:11], v[12:13], v[20:21], v[10:11]\n\t") "v_mov_b32 %0, v10" : "=v"(r32) :: FMA_CLOB);   // acc 2,3; sources both 0,1, 8 apart
        if constexpr (K == 14) asm volatile(FMA_INIT REP64("v_fma_f64 v[10:11], v[12:13], v[12:13], v[10:11]\n\t") "v_mov_b32 %0, v10" : "=v"(r32) :: FMA_CLOB);   // the two sources the same register
        if constexpr (K == 15) asm volatile(FMA_INIT REP16("v_fma_f64 v[10:11], v[12:13], v[16:17], v[10:11]\n\tv_fma_f64 v[10:11], v[14:15], v[20:21], v[10:11]\n\tv_fma_f64 v[10:11], v[16:17], v[18:19], v[10:11]\n\tv_fma_f64 v[10:11], v[12:13], v[14:15], v[10:11]\n\t") "v_mov_b32 %0, v10" : "=v"(r32) :: FMA_CLOB);
        // the step as the lab's kernel has it, on fixed registers: acc v[10:11]; everything else in banks 0,1 (K 16) or wherever (K 17: all 2,3)
        if constexpr (K == 16) asm volatile(FMA_INIT "s_mov_b64 vcc, 0\n\t" REP16(
            "v_cvt_f64_f32 v[24:25], v30\n\tv_cndmask_b32_dpp v31, v30, v32, vcc row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
            "v_fma_f64 v[10:11], v[12:13], v[16:17], v[10:11]\n\tv_fma_f64 v[10:11], v[20:21], v[28:29], v[10:11]\n\tv_fma_f64 v[10:11], v[24:25], v[36:37], v[10:11]\n\t"
            "v_fma_f64 v[10:11], v[40:41], v[44:45], v[10:11]\n\tv_cvt_f64_f32 v[20:21], v31\n\tv_cvt_f32_f64 v30, v[10:11]\n\tv_fma_f64 v[10:11], v[20:21], v[48:49], v[10:11]\n\t")
            "v_mov_b32 %0, v10" : "=v"(r32) :: FMA_CLOB, "vcc", "v24", "v25", "v28", "v29", "v30", "v31", "v32", "v36", "v37", "v40", "v41", "v44", "v45", "v48", "v49");
        if constexpr (K == 17) asm volatile(FMA_INIT "s_mov_b64 vcc, 0\n\t" REP16(
            "v_cvt_f64_f32 v[26:27], v30\n\tv_cndmask_b32_dpp v31, v30, v32, vcc row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
            "v_fma_f64 v[10:11], v[14:15], v[18:19], v[10:11]\n\tv_fma_f64 v[10:11], v[22:23], v[34:35], v[10:11]\n\tv_fma_f64 v[10:11], v[26:27], v[38:39], v[10:11]\n\t"
            "v_fma_f64 v[10:11], v[42:43], v[46:47], v[10:11]\n\tv_cvt_f64_f32 v[22:23], v31\n\tv_cvt_f32_f64 v30, v[10:11]\n\tv_fma_f64 v[10:11], v[22:23], v[50:51], v[10:11]\n\t")
            "v_mov_b32 %0, v10" : "=v"(r32) :: FMA_CLOB, "vcc", "v22", "v23", "v26", "v27", "v30", "v31", "v32", "v34", "v35", "v38", "v39", "v42", "v43", "v46", "v47", "v50", "v51");
    }
    asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (acc + r + r32 + (double)lacc == 123.456) t1 = 0;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + threadIdx.x] = acc + r + r32 + (double)lacc;
}
// what the hand-encoded v_cvt_f64_f32_dpp row_shr:1 returns per lane: old = -1.0, source = (float)lane
__global__ __launch_bounds__(64) void hack_probe(double *out)
{
    out[threadIdx.x] = cvt_dpp_shr1(-1.0, __float_as_uint((float)threadIdx.x + 0.25f));
}

template <int K> static void run_micro(const char *name)
{
    unsigned long long *d; double *sink; CHECK(hipMalloc(&d, 1024 * 8)); CHECK(hipMalloc(&sink, 1024 * 64 * 8));
    hipLaunchKernelGGL(micro<K>, dim3(1024), dim3(64), 0, 0, d, sink); CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(micro<K>, dim3(1024), dim3(64), 0, 0, d, sink); CHECK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(1024); CHECK(hipMemcpy(h.data(), d, 1024 * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double per = K >= 16 ? 144.0 : 64.0;
    printf("micro %-50s %6.2f cycles per instruction (median of 1024 waves; min %.2f)\n", name, h[512] / (32.0 * per), h[0] / (32.0 * per));
    hipFree(d); hipFree(sink);
}

static void cpu_ref(const std::vector<float> &coef, std::vector<int> &state, const std::vector<float> &in, std::vector<float> &out, int C, int B, int nsec, int c0, int c1)
{
    for (int c = c0; c < c1; c++)
        for (int n = 0; n < B; n++) {
            float xn = in[(size_t)n * C + c];
            for (int s = 0; s < nsec; s++) {
                const float *co = &coef[((size_t)c * 16 + s) * 5];
                int *st = &state[((size_t)c * 16 + s) * 6];
                double acc; long long bits = ((long long)st[1] << 32) | (unsigned)st[0]; memcpy(&acc, &bits, 8);
                float x1, x2, y1, y2; memcpy(&x1, &st[2], 4); memcpy(&x2, &st[3], 4); memcpy(&y1, &st[4], 4); memcpy(&y2, &st[5], 4);
                acc = fma((double)xn, (double)co[0], acc); acc = fma((double)x1, (double)co[1], acc); acc = fma((double)x2, (double)co[2], acc);
                acc = fma((double)y1, (double)co[3], acc); acc = fma((double)y2, (double)co[4], acc);
                const float yn = (float)acc;
                memcpy(&bits, &acc, 8); st[0] = (int)(unsigned)bits; st[1] = (int)(unsigned)((unsigned long long)bits >> 32);
                memcpy(&st[2], &xn, 4); memcpy(&st[3], &x1, 4); memcpy(&st[4], &yn, 4); memcpy(&st[5], &y1, 4);
                xn = yn;
            }
            out[(size_t)n * C + c] = xn;
        }
}

template <int VAR, int OM = 0>
static void run(const char *name, int C, int B, int nsec, const std::vector<float> &coef, const std::vector<float> &in, int check_ch)
{
    Args a{};
    float *d_coef, *d_in, *d_out; int *d_state;
    const size_t ncoef = (size_t)C * 16 * 5, nstate = (size_t)C * 16 * 6, nio = (size_t)B * C;
    CHECK(hipMalloc(&d_coef, ncoef * 4)); CHECK(hipMalloc(&d_state, nstate * 4)); CHECK(hipMalloc(&d_in, nio * 4)); CHECK(hipMalloc(&d_out, nio * 4));
    CHECK(hipMemcpy(d_coef, coef.data(), ncoef * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_in, in.data(), nio * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(d_state, 0, nstate * 4)); CHECK(hipMemset(d_out, 0, nio * 4));
    a.coef = d_coef; a.state = d_state; a.in = d_in; a.out = d_out; a.C = C; a.B = B; a.nsec = nsec;
    CHECK(hipMalloc(&a.stamps, (size_t)((C + 15) / 16) * 4 * 12 * 8));
    CHECK(hipMalloc(&a.out2, (size_t)C * (B + 64) * 4)); CHECK(hipMemset(a.out2, 0, (size_t)C * (B + 64) * 4));
    const int nblk = (C + 15) / 16;
    // correctness: two blocks in a row against the CPU loop on the first check_ch channels and the last 8
    std::vector<int> st_ref(nstate, 0); std::vector<float> out_ref(nio, 0.f), out(nio); std::vector<int> st(nstate);
    long long bad = 0, badst = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL((bq_lab<VAR, OM>), dim3(nblk), dim3(256), 0, 0, a);
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemcpy(out.data(), d_out, nio * 4, hipMemcpyDeviceToHost));
        cpu_ref(coef, st_ref, in, out_ref, C, B, nsec, 0, check_ch); cpu_ref(coef, st_ref, in, out_ref, C, B, nsec, std::max(check_ch, C - 8), C);
        if (OM >= 2 && OM != 6) {          // the steady batches' frames are in out2, the others in out: every frame must be in one of them
            std::vector<float> o2((size_t)C * (B + 64)); CHECK(hipMemcpy(o2.data(), a.out2, o2.size() * 4, hipMemcpyDeviceToHost));
            for (int n = 0; n < B; n++) for (int c = 0; c < C; c++) if (c < check_ch || c >= C - 8) {
                const float want = out_ref[(size_t)n * C + c], g1 = out[(size_t)n * C + c], g2 = o2[(size_t)c * (B + 64) + (OM == 5 ? 32 : 31) + n];
                bad += memcmp(&g1, &want, 4) != 0 && memcmp(&g2, &want, 4) != 0;
            }
            CHECK(hipMemset(d_out, 0, nio * 4)); CHECK(hipMemset(a.out2, 0, (size_t)C * (B + 64) * 4));
        } else
        for (int n = 0; n < B; n++) for (int c = 0; c < C; c++) if (c < check_ch || c >= C - 8) bad += memcmp(&out[(size_t)n * C + c], &out_ref[(size_t)n * C + c], 4) != 0;
    }
    CHECK(hipMemcpy(st.data(), d_state, nstate * 4, hipMemcpyDeviceToHost));
    for (int c = 0; c < C; c++) if (c < check_ch || c >= C - 8) for (int i = 0; i < nsec * 6; i++) badst += st[(size_t)c * 96 + i] != st_ref[(size_t)c * 96 + i];
    // timing
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    // the chip raises its clock over the first ~0.1 s of load: half a second of launches before the timed ones
    for (int i = 0; i < (C * B >= (1 << 20) ? 12000 : 3000); i++) hipLaunchKernelGGL((bq_lab<VAR, OM>), dim3(nblk), dim3(256), 0, 0, a);
    CHECK(hipEventRecord(e0));
    const int N = 1000;
    for (int i = 0; i < N; i++) hipLaunchKernelGGL((bq_lab<VAR, OM>), dim3(nblk), dim3(256), 0, 0, a);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s C=%5d B=%5d nsec=%2d  %8.2f us/launch (back to back)   mismatches: out %lld state %lld\n", name, C, B, nsec, ms * 1000.0 / N, bad, badst);
    if (VAR & 8) {
        const int nw = nblk * 4;
        std::vector<unsigned long long> h((size_t)nw * 4); CHECK(hipMemcpy(h.data(), a.stamps, h.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> cyc, clk;
        for (int w = 0; w < nw; w++) { const double dc = (double)(h[4 * w + 2] - h[4 * w]), dr = (double)(h[4 * w + 3] - h[4 * w + 1]); cyc.push_back(dc); clk.push_back(dc / dr * 0.1); }
        std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
        {
            std::vector<unsigned long long> hp((size_t)nw * 8); CHECK(hipMemcpy(hp.data(), a.stamps + (size_t)nblk * 16, hp.size() * 8, hipMemcpyDeviceToHost));
            printf("    phases (cycles since wave start, wave 0 / median wave): ");
            for (int i = 0; i < 5; i++) { std::vector<unsigned long long> v; for (int w = 0; w < nw; w++) v.push_back(hp[(size_t)w * 8 + i]); std::sort(v.begin(), v.end()); printf(" %llu/%llu", hp[i], v[nw / 2]); }
            printf("  [state loaded, first inputs staged, first six batches done, steady loop done, tail done]\n");
        }
        printf("    wave life: median %.0f cycles (min %.0f max %.0f) = %.1f cycles per step of %d; in-kernel clock median %.3f GHz (min %.3f max %.3f)\n",
               cyc[nw / 2], cyc[0], cyc[nw - 1], cyc[nw / 2] / (B + 2 * nsec - 1), B + 2 * nsec - 1, clk[nw / 2], clk[0], clk[nw - 1]);
    }
    hipFree(d_coef); hipFree(d_state); hipFree(d_in); hipFree(d_out);
}

int main()
{
    {
        double *d; CHECK(hipMalloc(&d, 64 * 8));
        hipLaunchKernelGGL(hack_probe, dim3(1), dim3(64), 0, 0, d); CHECK(hipDeviceSynchronize());
        double h[64]; CHECK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
        int ok = 1;
        for (int l = 0; l < 64; l++) ok &= (l & 15) ? h[l] == (double)((float)(l - 1) + 0.25f) : h[l] == -1.0;
        printf("hand-encoded v_cvt_f64_f32_dpp row_shr:1: %s  (lanes 0..17:", ok ? "WORKS" : "does NOT do a row shift");
        for (int l = 0; l < 18; l++) printf(" %g", h[l]);
        printf(")\n");
        hipFree(d);
    }
    run_micro<0>("v_cvt_f64_f32 (independent)");
    run_micro<1>("v_cvt_f64_f32_dpp row_shr:1 (hand-encoded)");
    run_micro<2>("v_mov_b32_dpp row_shr:1 (independent)");
    run_micro<3>("v_fma_f64 dependent chain");
    run_micro<4>("v_mad_i64_i32 dependent chain");
    run_micro<5>("v_cvt_f32_f64 -> v_cvt_f64_f32 dependent (per instr x2)");
    run_micro<6>("v_alignbit_b32 dependent");
    run_micro<7>("s_nop 0");
    run_micro<8>("s_mov_b64");
    run_micro<9>("v_mov_b64");
    run_micro<10>("fma: acc, a, b all in banks 2,3");
    run_micro<11>("fma: acc 2,3; a, b both 0,1");
    run_micro<12>("fma: acc 2,3; a 0,1; b 2,3");
    run_micro<13>("fma: acc 2,3; a, b both 0,1 (8 apart)");
    run_micro<14>("fma: a == b");
    run_micro<15>("fma x4: acc 2,3; sources mixed 0,1");
    run_micro<16>("9-instr step, acc 2,3, rest 0,1 (per instr x9)");
    run_micro<17>("9-instr step, everything 2,3 (per instr x9)");
    const int Cmax = 4096, Bmax = 2048;
    std::vector<float> coef((size_t)Cmax * 16 * 5), in((size_t)Bmax * Cmax);
    unsigned v = 12345;
    for (auto &x : in) { v = v * 1664525u + 1013904223u; x = (float)((int)v >> 3) / 2147483648.0f; }
    for (int c = 0; c < Cmax; c++)
        for (int s = 0; s < 16; s++) {
            // peaking EQ like bench.py's synthetic program
            const double f0 = 100 + 37 * s + 3 * (c % 97), Q = 0.7 + 0.05 * (s % 5), g = (s & 1) ? 1.2 : 0.8, fs = 48000;
            const double A = sqrt(g), w0 = 2 * M_PI * f0 / fs, al = sin(w0) / (2 * Q);
            const double a0 = 1 + al / A;
            float *co = &coef[((size_t)c * 16 + s) * 5];
            co[0] = (float)((1 + al * A) / a0); co[1] = (float)(-2 * cos(w0) / a0); co[2] = (float)((1 - al * A) / a0);
            co[3] = (float)(-(-2 * cos(w0) / a0) - 1.0); co[4] = (float)(-((1 - al / A) / a0));
        }
    auto sub = [&](int C, int B) { std::vector<float> r((size_t)B * C); for (int n = 0; n < B; n++) for (int c = 0; c < C; c++) r[(size_t)n * C + c] = in[(size_t)n * Cmax + c]; return r; };
    struct Cfg { int C, B, nsec; } cfgs[] = {{4096, 1024, 16}, {512, 1024, 16}, {4096, 2048, 16}, {2048, 1024, 8}, {37, 100, 5}, {16, 1, 16}, {20, 17, 1}};
    for (auto &cf : cfgs) {
        const auto x = sub(cf.C, cf.B);
        std::vector<float> co((size_t)cf.C * 16 * 5); memcpy(co.data(), coef.data(), co.size() * 4);
        const int chk = cf.C < 24 ? cf.C : 24;
        run<0, 6>("row_newbcast results (10 VALU)", cf.C, cf.B, cf.nsec, co, x, chk);
        run<0, 0>("results through LDS (9 VALU)", cf.C, cf.B, cf.nsec, co, x, chk);
        if (cf.C >= 512 && cf.nsec == 16) run<8, 6>("row_newbcast, clock stamps", cf.C, cf.B, cf.nsec, co, x, chk);
    }
    return 0;
}
