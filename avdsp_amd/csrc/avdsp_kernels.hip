/*
 * avdsp_kernels.hip -- hand-written gfx950 (MI355X / CDNA4) kernels for the AVDSP hot path and the
 * implementation of the thin C ABI in include/avdsp_hip.h.
 *
 * What runs here (reference file:line of the semantics each kernel reproduces):
 *   load stage      DSP_LOAD / DSP_LOAD_GAIN        dsp_runtime.c:565-607, dsp_fpmath.h:66-80,
 *                                                   dsp_ieee754.h:204-298,377-410
 *   biquad cascade  DSP_BIQUADS                     dsp_runtime.c:827-849, dsp_biquadSTD.h:25-119
 *   FIR             DSP_FIR (float accumulators)    dsp_runtime.c:928-969, dsp_firSTD.h:38-52
 *   store stage     DSP_SAT0DB / DSP_STORE          dsp_runtime.c:464-475,610-633, dsp_fpmath.h:84-98,
 *                                                   dsp_ieee754.h:85-107,187-199
 *
 * Data layout in HBM: the device keeps a word-for-word mirror of the caller's buffer (program words,
 * then the state area), so coefficient and state addresses are the program's own word offsets and a
 * checkpoint is a plain copy.  Samples are frame-interleaved [frame][channel] 32-bit words exactly
 * as the reference host hands them over (linux/avdsp_plugin.c:103-139).
 *
 * Kernels:
 *   biquad_pipe<FMT,P>  one lane per (channel, section): the cascade is a systolic pipeline across
 *                       the lanes of a 16-lane DPP row (section s works on frame t-s), which is
 *                       exact because every lane performs the reference's operations in the
 *                       reference's order; only independent work is overlapped.  State and
 *                       coefficients live in registers for the whole block.
 *   biquad_simple<FMT>  one lane per channel, sections in a loop, state in memory (cross-check path)
 *   fir_mfma<FMT,NG>    one workgroup per channel; taps and the input window staged in LDS; the
 *                       block's outputs are a dense (16 x K) x (K x 16) contraction per 256-frame
 *                       tile on v_mfma_f64_16x16x4_f64: Y[i][a] = sum_m h[m+i] * x[16a-m].
 *                       Products of two floats are exact in f64 and the K index ascends with the
 *                       tap index, i.e. the reference's summation order.
 *   fir_plain<FMT>      same staging, sequential v_fma_f64 tap loop per output (reference order)
 *   passthrough<FMT>    chains without filters
 */
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <type_traits>
#include <utility>
#include <vector>
#include <algorithm>

#include "avdsp_hip.h"
#include "avdsp_format.h"

namespace {

thread_local char g_err[512];
thread_local bool g_err_ready = false;   /* the latest set_err() was ready_check()'s (avdsp_hip_last_error_is_ready_timeout) */

int set_err(const char *fmt, ...)
{
    va_list ap;
    g_err_ready = false;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return -1;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return set_err("%s: %s", #expr, hipGetErrorString(e_));          \
    } while (0)

constexpr int kFirChunk   = 1024;     /* frames per launch: 4 MFMA tiles of 256 frames */
constexpr int kTileFrames = 256;
constexpr int kBlock      = 256;

/* ------------------------------------------------------------------------------------------
 * device helpers: arithmetic of the load / store stages
 * ---------------------------------------------------------------------------------------- */

/* The operand of the reference's float products (dspMulFloatDouble, dsp_ieee754.h:377-410), which are
 * assembled from the bit fields: a biased exponent of 0 (zero, subnormal) counts as +0 (:383-386), and
 * Inf / NaN are not recognised -- exponent 255 is read like any other, i.e. as 1.m x 2^128.        */
__device__ __forceinline__ float flush_f32(float v)
{
    return (__float_as_uint(v) & 0x7F800000u) ? v : 0.0f;
}
__device__ __forceinline__ double widen_exp255(unsigned u)              /* 1.m x 2^128 with the sign of u */
{
    const unsigned hi = (u & 0x80000000u) | (1151u << 20) | ((u & 0x7FFFFFu) >> 3);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | (unsigned long long)(u << 29)));
}
__device__ __forceinline__ double mulop(float v)
{
    const unsigned u = __float_as_uint(v);
    if ((u & 0x7F800000u) == 0x7F800000u) return widen_exp255(u);
    return (double)flush_f32(v);
}
/* Subnormal floats.  The reference runs with MXCSR.FTZ and MXCSR.DAZ set (it is built -Ofast; loading it
 * switches the thread): every SSE conversion and arithmetic instruction reads a subnormal float as signed
 * zero and flushes a subnormal result.  The wave's MODE register has the same two switches for single
 * precision, and the chain kernels turn them on at entry (tools/denorm_mode_probe.hip: v_cvt_f32_f64 then
 * flushes its result, v_cvt_f64_f32 its operand, signs kept).  (float)acc and (double)sample in those
 * kernels therefore ARE the reference's cvtsd2ss / cvtss2sd, at no instruction cost, and a filter decaying
 * into silence ends in the same signed zeros (oracle: ORC_FTZ_DAZ_ON, avdsp_oracle.c).  Double-precision
 * subnormals (< 2.2e-308) are out of reach.  The general interpreter keeps the default mode (its float
 * division relies on it) and flushes in software (avdsp_interp.inc).                                  */
__device__ __forceinline__ void flush_f32_subnormals_like_the_reference()
{
    __builtin_amdgcn_s_setreg(1 /* HW_REG_MODE */ | (4 << 6) /* FP_DENORM: single precision */ | (1 << 11) /* 2 bits */, 0);
}
__device__ __forceinline__ float  narrow_f32(double d) { return (float)d; }     /* under the flush MODE only */
__device__ __forceinline__ double widen_f32(float v)   { return (double)v; }
/* the same flush in software, for fir_mfma: there the MODE switch costs the MFMA loop 2.5 % (measured A/B,
 * reason unknown), while its few conversions -- one per output sample -- are off the critical path */
__device__ __forceinline__ unsigned ftz_bits(unsigned u) { return (u & 0x7F800000u) ? u : (u & 0x80000000u); }

/* dsp_ieee754.h:204-250: int -> float, magnitude TRUNCATED to 24 bits, times 2^-31.  INT_MIN leaves
 * the reference's 7-step normaliser one step short: mantissa 0, exponent 126, i.e. -0.5.       */
__device__ __forceinline__ float int_to_float_scaled31(int x)
{
    /* without a branch (round 5: the cascade's row kernel stages a sample per lane and batch through here, and three branches cost it more
     * than the conversion -- format 4's 4096 x 16 cascade in front of a FIR 13 us of 118): the magnitude normalised to bit 31, its top 24 bits
     * the mantissa (truncated), 127 - the leading zeros the exponent.  All 2^32 arguments compared with the branching form on the CPU. */
    const unsigned sign = (unsigned)x & 0x80000000u;
    const unsigned mag = x < 0 ? 0u - (unsigned)x : (unsigned)x;
    const unsigned lz = (unsigned)__clz((int)mag);                                   /* (32 for 0: the result is replaced below) */
    const unsigned norm = mag << (lz & 31u);
    unsigned r = sign | ((127u - lz) << 23) | ((norm >> 8) & 0x7FFFFFu);
    r = mag == 0x80000000u ? (sign | (126u << 23)) : r;
    r = x == 0 ? 0u : r;
    return __uint_as_float(r);
}

/* dsp_ieee754.h:253-298: exact x * 2^-31; INT_MIN mirrors the value the reference's -Ofast build
 * returns after negating INT_MIN (signed overflow): -2^-8.                                      */
__device__ __forceinline__ double int_to_double_scaled31(int x)
{
    if (x == (int)0x80000000u) return -0.00390625;
    return (double)x * 4.656612873077392578125e-10;      /* 2^-31, exact scaling */
}

/* dsp_ieee754.h:187-199 */
__device__ __forceinline__ double saturate_double_0db(double d)
{
    int e = (int)(__double_as_longlong(d) >> 52);
    if (e >= 1023) return 1.0;
    if (e < 0 && e >= -1025) return -1.0;
    return d;
}

/* dsp_ieee754.h:171-184, the float form of the same clamp: (float)saturate(acc) == saturate((float)acc) */
__device__ __forceinline__ float saturate_f32_0db(float f)
{
    const int e = (int)__float_as_uint(f) >> 23;               /* arithmetic shift: the sign stays on top */
    if (e >= 127) return 1.0f;
    if (e < 0 && e >= -129) return -1.0f;
    return f;
}

/* dsp_ieee754.h:85-107: truncate toward zero to s.31, |d| >= 1 -> +/-0x7FFFFFFF.  For |d| < 2^-42 the
 * reference shifts by >= 64 (undefined in C); its x86-64 binaries take the count modulo 64 and so
 * do the golden vectors, the oracle and this function.                                          */
__device__ __forceinline__ int s31_from_double(double d)
{
    long long u = __double_as_longlong(d);
    int e = (int)((u >> 52) & 2047);
    if (e == 0) return 0;
    long long m = (u & 0xFFFFFFFFFFFFFll) | (1ll << 52);
    int n = 1044 - e;
    if (n > 21) m >>= (n & 63);
    else m = 0x7FFFFFFF;
    if (u < 0) m = -m;
    return (int)m;
}

/* dsp_fpmath.h:84-98 with mant = 28 */
__device__ __forceinline__ long long saturate64_031(long long a)
{
    const long long pos = 1ll << 59;
    if (a >= pos) return 0x7FFFFFFFll;
    if (a < -pos) return (long long)0xFFFFFFFF80000000ull;
    return a >> 28;
}

/* Accumulator type per arithmetic model */
template <int FMT> struct Alu { using type = double; };
template <> struct Alu<2> { using type = long long; };

/* DSP_LOAD / DSP_LOAD_GAIN on one raw 32-bit sample word */
template <int FMT>
__device__ __forceinline__ typename Alu<FMT>::type load_stage(unsigned raw, int mode, unsigned gain_bits)
{
    if constexpr (FMT == 2) {
        long long s = (int)raw;
        return mode == AVDSP_LOAD_GAIN ? s * (long long)(int)gain_bits : s;          /* :571, :593 */
    } else if constexpr (FMT == 4) {
        if (mode == AVDSP_LOAD_GAIN)                                                  /* :596-598 */
            return (double)int_to_float_scaled31((int)raw) * mulop(__uint_as_float(gain_bits));      /* (the sample's float is zero or normal: mulop() of it is its widening) */
        return int_to_double_scaled31((int)raw);                                      /* :575 */
    } else {
        double x = widen_f32(__uint_as_float(raw));                                   /* :580, :603-604 */
        return mode == AVDSP_LOAD_GAIN ? x * widen_f32(__uint_as_float(gain_bits)) : x;
    }
}

/* What BIQUADS / FIR take from the accumulator: (int)(X >> 28) or (float)X */
template <int FMT>
__device__ __forceinline__ unsigned narrow_stage(typename Alu<FMT>::type X)
{
    if constexpr (FMT == 2) return (unsigned)(int)(X >> 28);                          /* dspShiftInt, :831 */
    else return __float_as_uint(narrow_f32(X));
}

/* store_stage<4> without a branch (round 5; biquad_row's ACC form converts once per 16 steps, every lane, and three taken branches there cost as
 * much as the conversion; the FIR epilogues convert a word per sample): [saturate_double_0db], s31_from_double -- whose int result is the low word of +-(m >> (n & 63)) --, the dither mask */
__device__ __forceinline__ unsigned store_word_f4(double X, bool sat, int mask)
{
    long long u = __double_as_longlong(X);
    const int se = (int)(u >> 52);                                                    /* sign and exponent, sign-extended */
    const bool up = sat && se >= 1023, down = sat && se < 0 && se >= -1025;
    u = up ? 0x3FF0000000000000ll : down ? (long long)0xBFF0000000000000ull : u;
    const int e = (int)((u >> 52) & 2047);
    const unsigned long long m = ((unsigned long long)u & 0xFFFFFFFFFFFFFull) | (1ull << 52);
    const int n = 1044 - e;
    unsigned w = (unsigned)(m >> (n & 63));
    w = n > 21 ? w : 0x7FFFFFFFu;
    w = e == 0 ? 0u : w;
    w = u < 0 ? 0u - w : w;
    return w & (unsigned)mask;
}

/* [DSP_SAT0DB] + DSP_STORE -> raw 32-bit sample word */
template <int FMT>
__device__ __forceinline__ unsigned store_stage(typename Alu<FMT>::type X, int sat, int mask)
{
    if constexpr (FMT == 2) {
        if (sat) X = saturate64_031(X);
        return (unsigned)((int)X & mask);                                             /* :616-618 */
    } else {
        if constexpr (FMT == 4) return store_word_f4(X, sat == 1, mask);              /* :622-627 (saturate_double_0db, s31_from_double, the mask) */
        if (sat) X = saturate_double_0db(X);
        return __float_as_uint(narrow_f32(X));                                        /* :629-630 */
    }
}

/* XCD-aware block index: hardware deals consecutive workgroups round-robin over the 8 XCDs, so
 * giving XCD x the x-th contiguous eighth of the work keeps neighbouring channels (which share
 * 128-byte lines of the interleaved sample block) in the same L2.  Speed only, never correctness. */
__device__ __forceinline__ int xcd_remap(int b, int per_xcd) { return (b & 7) * per_xcd + (b >> 3); }

/* FIR history, device-native: one ring of R floats (R a power of two) per chain.  Frame n of the block
 * being processed lives at (wpos + n) & (R-1); older samples sit behind it.  The reference layout of
 * the delay line (st[i] = x[n-1-i], dsp_firSTD.h:45-50) is produced from the ring only when the
 * host asks for the state (ring_to_state) and loaded back by state_to_ring.                     */
struct Ring {
    float *base;        /* rows of 2 R floats: every sample is stored TWICE, at p and at p + R (round 4), so that a reader may run up to R
                         * floats on from any position without wrapping -- fir_tile requests a chunk's window with ONE masked offset per
                         * lane and immediates from there (a masked offset per sample was a third of a chunk boundary's instructions) */
    int    R;
    int    wpos;
    double *wide;       /* the same ring as the FIR's window operand: mulop(sample), doubles, frame n at (wpos + n + 3) & (R-1); rows of 2 R
                         * doubles, every operand twice like the floats (fir_flow copies a chunk's window from one masked start) */
};
__device__ __forceinline__ float *ring_row(const Ring &r, int cid) { return r.base + (size_t)cid * 2 * r.R; }
__device__ __forceinline__ double *wide_row(const Ring &r, int cid) { return r.wide + (size_t)cid * 2 * r.R; }
__device__ __forceinline__ float *ring_at(const Ring &r, int cid, int q)
{
    return ring_row(r, cid) + ((r.wpos + q) & (r.R - 1));
}
/* every writer of the ring goes through here: the float (the reference's delay-line value) and the operand made of it */
/* wt: write-through stores (`sc1`) -- what a cascade puts into the ring of a launch whose FIR waits for the chains' ready words
 * (below) is handed to waves of ANOTHER launch, and write-through payload stores are that hand-over's cheap form: no release fence,
 * which would write back the whole XCD's L2 under the running FIR (MI355X_MICROARCH.md, "Valid forms", R1).  Everything else stores
 * the plain way (a write-through store drops its line from the L2: the FIR-only chains' own appends are read back at once). */
__device__ __forceinline__ void ring_put(const Ring &r, int cid, int q, unsigned bits, bool wt = false)
{
    float *at = ring_at(r, cid, q);
    double *wat = r.wide ? wide_row(r, cid) + ((r.wpos + q + 3) & (r.R - 1)) : nullptr;
    if (wt) {
        __hip_atomic_store(reinterpret_cast<unsigned *>(at), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<unsigned *>(at + r.R), bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wat) {
            const unsigned long long w = (unsigned long long)__double_as_longlong(mulop(__uint_as_float(bits)));
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(wat), w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(wat + r.R), w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    at[0] = __uint_as_float(bits); at[r.R] = __uint_as_float(bits);
    if (wat) { const double w = mulop(__uint_as_float(bits)); wat[0] = w; wat[r.R] = w; }
}

/* Cascade -> FIR inside the device, without an event between two queues ("ready words"): the cascade of launch n
 * leaves `n` in ready[chain] once the chain's block is in the ring, the FIR wave of launch n that takes the chain waits for it.
 * The hand-over is MI355X_MICROARCH.md's R1: the ring stores are write-through (`sc1`, ring_put), every storing wave drains them
 * (s_waitcnt vmcnt(0)), then a relaxed agent-scope store of the word; the reader polls it relaxed (one word, all lanes the same
 * address), then one agent-scope acquire, then plain loads.  (A first version released with an agent-scope fence per cascade wave:
 * buffer_wbl2 writes back the XCD's whole L2, under the FIR that is filling it -- the cascade alone went 34 -> 52 us, the 4096-chain
 * step 0.519 -> 0.541 ms.)  The numbers only grow (compared modulo 2^32), so nothing is re-armed between launches.  The poll is
 * bounded: the cascade it waits for was enqueued before the FIR and needs nothing of it, so the bound is never met; if it ever
 * were (a preempted or shared GPU, a copy the cascade waits for that never comes), the wave leaves a mark and goes on rather than
 * hang the device, and the mark turns every later call into an error (ready_check below; round 5).                            */
__device__ __forceinline__ void chain_ready_release()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      /* every storing wave drains its write-through ring stores (ring_put) before its chains' words go out */
}
__device__ __forceinline__ void chain_ready_publish(unsigned *ready, int cid, unsigned seq)
{
    __hip_atomic_store(ready + cid, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void chain_ready_wait(unsigned *ready, int cid, unsigned seq, unsigned *timeouts, bool acquire = true)
{
    unsigned spins = 0;
    while ((int)(__hip_atomic_load(ready + cid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - seq) < 0) {
        __builtin_amdgcn_s_sleep(32);
        if (++spins > (1u << 19)) {                        /* ~ half a second */
            /* The wave goes on (it must end) with whatever the ring holds, so the block is NOT the reference's bits -- and the host
             * must learn of it without asking: timeouts[0] counts such waves, timeouts[2..3] is the device address of a word in
             * mapped pinned HOST memory, which every later entry point of the library looks at first (ready_check: the call then
             * fails with a negative code and a text, like every other failure of this library). */
            if ((threadIdx.x & 63) == 0) {
                atomicAdd(timeouts, 1u);
                unsigned *flag = *reinterpret_cast<unsigned *const *>(timeouts + 2);
                if (flag) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            break;
        }
    }
    /* (the acquire invalidates the XCD's L2 for every wave that passes here -- 20 us on every launch, measured.  Where the word was
     * set by a kernel BEHIND the cascade, nothing of the block can be in this XCD's L2 from before the cascade's write-back: no wave
     * reads a chain's new ring positions before it has seen the chain's word, and the launch began with an invalidate of its own) */
    if (acquire) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

struct BlockIO {
    const unsigned *in;  int in_stride,  in_base;
    unsigned       *out; int out_stride, out_base;
    int nframes;
    int store_mask;
};

/* final stores of one chain for frame n */
__device__ __forceinline__ void emit_out(const BlockIO &io, const avdsp_chain &c, int n, unsigned word)
{
#pragma unroll
    for (int k = 0; k < AVDSP_MAX_STORES; k++)          /* static indices keep the chain record in registers */
        if (k < c.n_out) io.out[(size_t)n * io.out_stride + (c.out_io[k] - io.out_base)] = word;
}

/* ------------------------------------------------------------------------------------------
 * biquad cascade, section-pipelined
 *
 * One lane per (channel, section); P = lanes reserved per channel (power of two >= sections).
 * At step t section s works on frame t - s, so after the pipeline has filled every lane runs one
 * biquad update per step and hands its result to lane+1 with a single DPP row shift.  Each lane
 * performs exactly the reference's operations in the reference's order (dsp_biquadSTD.h:37-74,
 * 87-117); only independent (channel, section, frame) triples overlap.
 *
 * Sample IO is batched so that the step loop contains no memory instruction:
 *   input   every NB = min(P,16) steps the first NB lanes of a channel load NB consecutive frames
 *           (three batches ahead), convert them once, and the batch is rotated one lane per step
 *           (DPP row_ror:15) so the section-0 lane always finds "its" frame in its own register;
 *   output  the last section drops its accumulator into a register that rotates the other way
 *           (row_ror:1); after NB steps the NB results sit in NB different lanes, which convert
 *           (SAT0DB/STORE) and store them together.
 * ---------------------------------------------------------------------------------------- */
/* what biquad_row needs of a chain and of a section, one record per row slot / per lane of a launch group; made with the plan */
struct RowRec { int cid, in_io, out_io, flags; unsigned gain_bits; int pad[3]; };      /* flags: load_mode | sat << 8 | to_ring << 9 | n_out << 16; pad[0]: the row's section count
                                                                                            (a launch with BiquadArgs::nsec 0 holds rows of several counts, four-row waves of one
                                                                                            count each; cid -1: a row that only fills its wave) */
struct LaneRec { int coef_word, state_word; };                                          /* -1: the lane holds no section */

struct BiquadArgs {
    int            *buf;            /* device mirror of the caller's buffer */
    const RowRec   *rows;           /* biquad_row: [ngroup] */
    const LaneRec  *lanes;          /* biquad_row: [ngroup][16], sections right-aligned in the row */
    const avdsp_chain *chains;
    const int      *sec_coef, *sec_state;
    const int      *group;          /* chain ids handled by this launch (all with `nsec` sections) */
    int             ngroup;
    int             nsec;
    Ring            ring;           /* where (float)X goes when a FIR follows the cascade */
    int             per_xcd;
    BlockIO         io;
    unsigned       *ready;          /* [chain]: number of the latest launch whose cascade has left this chain's block in the ring (chain_ready below); may be null */
    unsigned        seq;            /* ... this launch's number */
#ifdef AVDSP_BQ_STAMPS
    unsigned long long *stamps;     /* diagnostic build (tools/cascade_timeline.py): 32 s_memtime stamps per wave */
#endif
};
#ifdef AVDSP_BQ_STAMPS
#define BQ_STAMP(i) do { if ((threadIdx.x & 63) == 0 && (i) < 32) a.stamps[(size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BQ_STAMP(i) do { } while (0)
#endif

template <int CTRL>
__device__ __forceinline__ unsigned dpp_mov(unsigned old, unsigned src)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, 0xF, 0xF, false);
}
constexpr int kRowShr1 = 0x111, kRowRor1 = 0x121, kRowRor15 = 0x12F;

/* value of lane-1 (lane 0 of a 16-lane row, which has no source, keeps `old`) */
template <int P>
__device__ __forceinline__ unsigned from_prev_lane(unsigned old, unsigned v)
{
    if constexpr (P <= 16) return dpp_mov<kRowShr1>(old, v);
    else return (unsigned)__shfl_up((int)v, 1, 64);
}

/* what travels from section to section (and down the input batch): the 32-bit result only -- (int)(acc >> 28)
 * or the bits of (float)acc.  The float models widen it where it is used: one v_cvt_f64_f32 in the receiving
 * lane is cheaper than moving the double's two halves through DPP as well (8 -> 4 DPP moves per step). */
template <int FMT> struct Hand { unsigned y; };

template <int FMT>
__device__ __forceinline__ Hand<FMT> hand_from_sample(unsigned raw, int load_mode, unsigned gain_bits)
{
    Hand<FMT> h;
    h.y = narrow_stage<FMT>(load_stage<FMT>(raw, load_mode, gain_bits));
    return h;
}
/* Pieces of a long cascade (round 5; more than 64 sections: add_plan cuts the chain into pieces of up to 64 that run one launch after the
 * other).  What travels between two sections is a 32-bit word -- (int)(acc >> 28), or the bits of (float)acc -- and between two PIECES the
 * same word goes through a scratch column: kLoadRaw takes a sample word as the first section's input as it is, kStoreRaw (in the chain
 * record's `sat`) stores the last section's result word as it is.  Device-side only; the host's descriptors never hold them. */
constexpr int kLoadRaw = 2, kStoreRaw = 2;
template <int FMT>
__device__ __forceinline__ Hand<FMT> hand_from_sample_or_raw(unsigned raw, int load_mode, unsigned gain_bits)
{
    Hand<FMT> h = hand_from_sample<FMT>(raw, load_mode, gain_bits);
    if (load_mode == kLoadRaw) h.y = raw;
    return h;
}

template <int FMT, int CTRL>
__device__ __forceinline__ Hand<FMT> hand_rotate(Hand<FMT> h)
{
    h.y = dpp_mov<CTRL>(h.y, h.y);
    return h;
}

/* One chain's cascade over the block in the reference's own order -- frames outer, sections inner, state in memory
 * (dsp_biquadSTD.h:37-74, 87-117), every product through mulop(), i.e. with the bit-field reading of exponent 255.
 * Slow and obviously sequential: biquad_simple runs it for every chain (cross-check path; until round 5 cascades longer than 64
 * sections), biquad_pipe for a chain whose block turned up an Inf or NaN.                                        */
template <int FMT>
__device__ void cascade_in_reference_order(const BiquadArgs &a, int cid, const avdsp_chain &c)
{
    using alu_t = typename Alu<FMT>::type;
    const unsigned *inp = a.io.in + (c.in_io - a.io.in_base);
    for (int n = 0; n < a.io.nframes; n++) {
        const unsigned raw_in = inp[(size_t)n * a.io.in_stride];
        alu_t X = load_stage<FMT>(raw_in, c.load_mode, c.gain_bits);
        unsigned xin = c.load_mode == kLoadRaw ? raw_in : narrow_stage<FMT>(X);
        for (int s = 0; s < c.nsec; s++) {
            const int *co = a.buf + a.sec_coef[c.sec_base + s];
            int *st = a.buf + a.sec_state[c.sec_base + s];
            const unsigned long long raw = ((unsigned long long)(unsigned)st[1] << 32) | (unsigned)st[0];
            const unsigned x1 = (unsigned)st[2], x2 = (unsigned)st[3], y1 = (unsigned)st[4], y2 = (unsigned)st[5];
            unsigned yn;
            unsigned long long keep;
            if constexpr (FMT == 2) {
                unsigned long long u = raw;
                u += (unsigned long long)((long long)(int)xin * co[0]);
                u += (unsigned long long)((long long)(int)x1 * co[1]);
                u += (unsigned long long)((long long)(int)x2 * co[2]);
                u += (unsigned long long)((long long)(int)y1 * co[3]);
                u += (unsigned long long)((long long)(int)y2 * co[4]);
                long long acc = (long long)u;
                const int hi = (int)(acc >> 32);
                if (hi >= (1 << 27)) acc = (1ll << 59) - 1;
                else if (hi <= 1 - (1 << 27)) acc = -(1ll << 59);
                X = acc; keep = (unsigned long long)acc;
                yn = (unsigned)(int)(acc >> 28);
            } else {
                double acc = __longlong_as_double((long long)raw);
                acc = __builtin_fma(mulop(__uint_as_float(xin)), mulop(__int_as_float(co[0])), acc);
                acc = __builtin_fma(mulop(__uint_as_float(x1)), mulop(__int_as_float(co[1])), acc);
                acc = __builtin_fma(mulop(__uint_as_float(x2)), mulop(__int_as_float(co[2])), acc);
                acc = __builtin_fma(mulop(__uint_as_float(y1)), mulop(__int_as_float(co[3])), acc);
                acc = __builtin_fma(mulop(__uint_as_float(y2)), mulop(__int_as_float(co[4])), acc);
                X = acc; keep = (unsigned long long)__double_as_longlong(acc);
                yn = __float_as_uint(narrow_f32(acc));
            }
            st[0] = (int)(unsigned)keep; st[1] = (int)(unsigned)(keep >> 32);
            st[2] = (int)xin; st[3] = (int)x1; st[4] = (int)yn; st[5] = (int)y1;
            xin = yn;
        }
        if (c.fir_taps) ring_put(a.ring, cid, n, narrow_stage<FMT>(X), a.ready != nullptr);
        else if (c.sat == kStoreRaw) emit_out(a.io, c, n, xin);      /* (a piece of a long cascade: the last section's result word) */
        else emit_out(a.io, c, n, store_stage<FMT>(X, c.sat, a.io.store_mask));
    }
}

template <int FMT, int P>
__global__ __launch_bounds__(kBlock) void biquad_pipe(const BiquadArgs a)
{
    if constexpr (FMT != 2) flush_f32_subnormals_like_the_reference();
    BQ_STAMP(0);
    using alu_t = typename Alu<FMT>::type;
    constexpr int NB = P < 16 ? P : 16;                 /* steps per IO batch */
    constexpr int CPB = kBlock / P;                       /* chains per block */
    constexpr int DEPTH = 3;                            /* input batches in flight (the batch loop is unrolled by it) */
    static_assert(DEPTH == 3, "kNext below is written out for three slots");
    const int tid = threadIdx.x, rowpos = tid & 15;
    const int blk = xcd_remap(blockIdx.x, a.per_xcd);
    const int slot = blk * CPB + tid / P;
    const int s = tid % P;
    const Ring ring_l = a.ring;                         /* by value: the lambdas below must not make the kernel arguments addressable */
    const BlockIO io_l = a.io;
    const int nsec = a.nsec, B = io_l.nframes;
    const bool have_chain = slot < a.ngroup;
    /* (lanes without a chain fetch like the others -- see below -- so they take the launch's FIRST chain, whose input column the launch's
     * block has: chain 0 of the plan may be a piece of a long cascade with a column in another launch's scratch block) */
    const int cid = a.group[have_chain ? slot : 0];
    const avdsp_chain c = a.chains[cid];
    const bool lane_on = have_chain && s < nsec;
    const bool first = s == 0;
    const bool last = lane_on && s == nsec - 1;

    /* which (chain, step-in-batch) this lane ends up holding in the output batch register */
    int d, oslot;
    bool owner;
    if constexpr (P >= 16) {
        d = (rowpos - ((nsec - 1) & 15)) & 15;
        owner = (s >> 4) == ((nsec - 1) >> 4);
        oslot = slot;
    } else {
        d = (rowpos - (nsec - 1)) & (P - 1);
        owner = true;
        const int src = (rowpos - d) & 15;               /* row position of the lane that inserted the value */
        oslot = blk * CPB + (tid >> 4) * (16 / P) + (((src - (nsec - 1)) & 15) / P);
    }
    owner = owner && oslot < a.ngroup;
    const int ocid = a.group[owner ? oslot : 0];
    const avdsp_chain oc = a.chains[ocid];
    const int ostep = NB - 1 - d;                       /* step of the batch whose result this lane stores */
    [[maybe_unused]] const unsigned long long lastmask = __ballot(last);   /* lanes whose results leave the cascade */

    /* coefficients and the 6 state words of this (chain, section) stay in registers */
    int sw = 0;
    alu_t acc = 0;
    unsigned x1 = 0, x2 = 0, y1 = 0, y2 = 0;            /* raw state words: int32 or float bits */
    [[maybe_unused]] int    ci[5] = {0, 0, 0, 0, 0};
    [[maybe_unused]] double cd[5] = {0, 0, 0, 0, 0};
    [[maybe_unused]] double dx1 = 0, dx2 = 0, dy1 = 0, dy2 = 0;
    if (lane_on) {
        const int cw = a.sec_coef[c.sec_base + s];
        sw = a.sec_state[c.sec_base + s];
        const int *st = a.buf + sw;
        const unsigned long long raw = ((unsigned long long)(unsigned)st[1] << 32) | (unsigned)st[0];
        if constexpr (FMT == 2) {
            acc = (long long)raw;
            for (int k = 0; k < 5; k++) ci[k] = a.buf[cw + k];
        } else {
            acc = __longlong_as_double((long long)raw);
            for (int k = 0; k < 5; k++) cd[k] = mulop(__int_as_float(a.buf[cw + k]));
        }
        x1 = (unsigned)st[2]; x2 = (unsigned)st[3]; y1 = (unsigned)st[4]; y2 = (unsigned)st[5];
        if constexpr (FMT != 2) {
            dx1 = mulop(__uint_as_float(x1)); dx2 = mulop(__uint_as_float(x2));
            dy1 = mulop(__uint_as_float(y1)); dy2 = mulop(__uint_as_float(y2));
        }
    }

    /* input batches: lane s < NB of a channel fetches frame (batch*NB + s), clamped into the block.  Every lane loads
     * (the others a word of their channel nobody looks at): a load under a branch would hide from the compiler how many
     * memory operations follow it, and it would then wait for ALL of them before each batch (s_waitcnt vmcnt(0)). */
    const unsigned *inp = io_l.in + (c.in_io - io_l.in_base);
    auto fetch = [&](int batch) __attribute__((always_inline)) -> unsigned {
        int n = batch * NB + (s & (NB - 1));
        n = n < B ? n : B - 1;
        return inp[(size_t)n * io_l.in_stride];
    };
    unsigned rawq[DEPTH];
#pragma unroll
    for (int k = 0; k < DEPTH; k++) rawq[k] = fetch(k);

    Hand<FMT> hy;                                       /* this lane's latest result, offered to lane+1 */
    hy.y = y1;
    Hand<FMT> xcur;                                     /* operand of this step, fetched during the previous one */
    xcur.y = 0;
    unsigned ob_lo = 0, ob_hi = 0;                      /* rotating output batch (accumulator bits) */

    /* Step u does two things that do not depend on each other.  FETCH: take the operand of the NEXT step -- the
     * previous section's latest result through DPP (it is the result of step u-1, i.e. of compute index u-2), or the
     * next input sample for section 0.  COMPUTE index u-1 with the operand fetched one step ago.  So section s works
     * on frame n = (u-1) - 2s: the hand-off skews the pipeline by TWO steps per section, not one, and its
     * DPP move and widening conversion lie a whole step ahead of their use instead of at the head of the dependent
     * chain -- the loop-carried chain of a step is then the five accumulations alone (it was those five plus
     * cvt, DPP, cvt: 132 cycles per step with one wave per SIMD).  Costs nsec-1 more steps of fill per block.   */
    auto step = [&](Hand<FMT> &ib, int u, auto masked) __attribute__((always_inline)) {
        const Hand<FMT> ibr = hand_rotate<FMT, kRowRor15>(ib);
        Hand<FMT> xnext;
        xnext.y = from_prev_lane<P>(ib.y, hy.y);
        if constexpr (P != 16) {                        /* section-0 lanes that do not sit at a row start */
            if (first) xnext = ib;
        }
        ib = ibr;
        ob_lo = dpp_mov<kRowRor1>(ob_lo, ob_lo);
        if constexpr (FMT == 4) ob_hi = dpp_mov<kRowRor1>(ob_hi, ob_hi);
        bool act = true;
        if constexpr (decltype(masked)::value) { const int n = u - 1 - 2 * s; act = lane_on && n >= 0 && n < B; }
        if (act) {
            if constexpr (FMT == 2) {
                /* dsp_biquadSTD.h:37-74: five 32x32 MACs onto the previous full-precision output,
                 * saturate on the high word, keep acc, y = acc >> 28                              */
                unsigned long long v = (unsigned long long)acc;
                v += (unsigned long long)((long long)(int)xcur.y * ci[0]);
                v += (unsigned long long)((long long)(int)x1 * ci[1]);
                v += (unsigned long long)((long long)(int)x2 * ci[2]);
                v += (unsigned long long)((long long)(int)y1 * ci[3]);
                v += (unsigned long long)((long long)(int)y2 * ci[4]);
                acc = (long long)v;
                const int hi = (int)(acc >> 32);
                if (hi >= (1 << 27)) acc = (1ll << 59) - 1;
                else if (hi <= 1 - (1 << 27)) acc = -(1ll << 59);
                x2 = x1; x1 = xcur.y; y2 = y1;
                y1 = (unsigned)(int)(acc >> 28);
                hy.y = y1;
            } else {
                /* dsp_biquadSTD.h:87-117: exact float x float products, five sequential f64 adds */
                const double dxin = (double)__uint_as_float(xcur.y);    /* flushed by the MODE like mulop's exponent-0 rule */
                acc = __builtin_fma(dxin, cd[0], acc);
                acc = __builtin_fma(dx1, cd[1], acc);
                acc = __builtin_fma(dx2, cd[2], acc);
                acc = __builtin_fma(dy1, cd[3], acc);
                acc = __builtin_fma(dy2, cd[4], acc);
                /* (float)acc flushed like the reference's cvtsd2ss under FTZ; its product operand is then simply
                 * the widened value (+-0.0 adds nothing, like mulop's +0.0).  NOT mirrored in this loop: an Inf / NaN
                 * travelling down the cascade.  The reference's bit-field product reads exponent 255 as
                 * 1.m x 2^128 (mulop); honouring that for every section input and output costs this kernel
                 * 7 % (measured) for samples no audio stream contains.  Such a block is caught instead: see `odd`. */
                const float yn = narrow_f32(acc);
                x2 = x1; x1 = xcur.y; y2 = y1; y1 = __float_as_uint(yn);
                dx2 = dx1; dx1 = dxin; dy2 = dy1; dy1 = (double)yn;
                hy.y = y1;
            }
            if (last) {
                /* What leaves the cascade.  Only DSP_FORMAT 4 needs the whole accumulator later (31 bits of it go
                 * into the int sample); the float sample of format 6 is a function of (float)acc alone -- SAT0DB
                 * before or after the rounding gives the same float -- and the int64 store is either the low word
                 * of acc or, behind SAT0DB, acc >> 28, which the section has just formed.                     */
                if constexpr (FMT == 4) {
                    const unsigned long long bits = (unsigned long long)__double_as_longlong(acc);
                    ob_lo = (unsigned)bits; ob_hi = (unsigned)(bits >> 32);
                } else if constexpr (FMT == 6) ob_lo = y1;
                else ob_lo = c.sat ? y1 : (unsigned)(unsigned long long)acc;
            }
        }
        xcur = xnext;
    };

    const int steps = B + 2 * nsec - 1;                 /* u = 0 .. B + 2 (nsec - 1) */
    const int nbatches = (steps + NB - 1) / NB;
    /* The batch loop is unrolled DEPTH times so that every queue slot is a register of its own: a queue that shifts
     * (rawq[k] = rawq[k+1]) moves registers that loads are still in flight to, which costs a wait for the YOUNGEST
     * load at every batch -- the prefetch distance was one batch, not three, and a third of the kernel's time was
     * s_waitcnt (profiles/r01_cfg3: SQ_WAIT_ANY 33 %). */
    const int c_load_mode = c.load_mode; const unsigned c_gain_bits = c.gain_bits;      /* (scalars: the chain record itself stays out of the lambdas) */
    Hand<FMT> ib = hand_from_sample_or_raw<FMT>(rawq[0], c_load_mode, c_gain_bits);
    rawq[0] = fetch(DEPTH);
    BQ_STAMP(1);
    /* the NB steps of a batch in which every lane is busy */
    auto fast_steps = [&](int tb) __attribute__((always_inline)) {
        if constexpr (FMT == 6 && P == 16) {
                    /* The same thirteen instructions per step, in an order the compiler does not find: it issues the
                     * five dependent v_fma_f64 back to back (each waits ~4 cycles for its predecessor) and the eight
                     * independent instructions after them.  Here one of those sits behind every FMA, and the first FMA of
                     * the NEXT step (acc + x*b0: its operand was fetched a step ago) is issued under this step's
                     * (float)acc, so the conversions are off the chain as well.  accN = acc + x*b0 enters and leaves.  */
                    double accN = __builtin_fma((double)__uint_as_float(xcur.y), cd[0], acc);
                    double dxc = (double)__uint_as_float(xcur.y);
                    unsigned xb = xcur.y, ibv = ib.y;
#pragma unroll
                    for (int i = 0; i < NB; i++) {
                        double accQ, dyn, dxn;
                        unsigned yb, ibr;
                        asm volatile(
                            "v_fma_f64 %[an], %[dx1], %[c1], %[an]\n\t"
                            "v_mov_b32_dpp %[ibr], %[ib] row_ror:15 row_mask:0xf bank_mask:0xf\n\t"
                            "v_fma_f64 %[an], %[dx2], %[c2], %[an]\n\t"
                            "v_mov_b32_dpp %[ib], %[hy] row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                            "v_fma_f64 %[an], %[dy1], %[c3], %[an]\n\t"
                            "v_mov_b32_dpp %[ob], %[ob] row_ror:1 row_mask:0xf bank_mask:0xf\n\t"
                            "v_fma_f64 %[an], %[dy2], %[c4], %[an]\n\t"
                            "v_cvt_f64_f32 %[dxn], %[ib]\n\t"
                            "v_cvt_f32_f64 %[yb], %[an]\n\t"
                            "v_fma_f64 %[aq], %[dxn], %[c0], %[an]\n\t"
                            "v_cndmask_b32 %[ob], %[ob], %[yb], %[lm]\n\t"
                            "v_cvt_f64_f32 %[dyn], %[yb]\n\t"
                            : [an] "+v"(accN), [ib] "+v"(ibv), [ob] "+v"(ob_lo), [aq] "=&v"(accQ), [yb] "=&v"(yb), [dyn] "=&v"(dyn),
                              [dxn] "=&v"(dxn), [ibr] "=&v"(ibr)
                            : [dx1] "v"(dx1), [dx2] "v"(dx2), [dy1] "v"(dy1), [dy2] "v"(dy2), [c0] "v"(cd[0]), [c1] "v"(cd[1]),
                              [c2] "v"(cd[2]), [c3] "v"(cd[3]), [c4] "v"(cd[4]), [hy] "v"(hy.y), [lm] "s"(lastmask));
                        acc = accN;                      /* the finished accumulator of this step */
                        x2 = x1; x1 = xb; y2 = y1; y1 = yb;
                        dx2 = dx1; dx1 = dxc; dy2 = dy1; dy1 = dyn;
                        hy.y = yb;
                        xb = ibv; dxc = dxn;             /* operand of the next step: fetched and widened above */
                        ibv = ibr;
                        accN = accQ;
                    }
                    xcur.y = xb;
                    ib.y = ibv;
                } else {
#pragma unroll
                    for (int i = 0; i < NB; i++) step(ib, tb + i, std::false_type{});
                }
    };
    /* what a lane's output-batch register holds, as the word that goes to the ring (a FIR follows) or to the output block */
    auto flush_word = [&](bool to_ring) __attribute__((always_inline)) -> unsigned {
        if constexpr (FMT == 4) {
            const double X = __longlong_as_double((long long)(((unsigned long long)ob_hi << 32) | ob_lo));
            return to_ring || oc.sat == kStoreRaw ? narrow_stage<FMT>(X) : store_stage<FMT>(X, oc.sat, io_l.store_mask);
        } else if constexpr (FMT == 6) {
            return to_ring ? ob_lo : (oc.sat == 1 ? __float_as_uint(saturate_f32_0db(__uint_as_float(ob_lo))) : ob_lo);
        } else
            return oc.sat == kStoreRaw ? ob_lo : ob_lo & (unsigned)io_l.store_mask;      /* (kStoreRaw: ob_lo is y1, like behind SAT0DB) */
    };
    constexpr int kNext[3] = {1 % DEPTH, 2 % DEPTH, 0};
    /* a batch anywhere in the block: steps masked where the pipeline fills or drains, fetch clamped into the block, flush checked */
    auto batch_any = [&](int b, auto jc) __attribute__((always_inline)) {
        constexpr int j = decltype(jc)::value;
        const int tb = b * NB;
        if (tb >= 2 * nsec - 1 && tb + NB <= B + 1) fast_steps(tb);
        else {
#pragma unroll
            for (int i = 0; i < NB; i++) step(ib, tb + i, std::true_type{});
        }
        /* batch b+1's samples were requested DEPTH batches ago; its slot is then refilled for batch b+1+DEPTH */
        ib = hand_from_sample_or_raw<FMT>(rawq[kNext[j]], c_load_mode, c_gain_bits);
        rawq[kNext[j]] = fetch(b + 1 + DEPTH);
        /* flush the output batch: this lane holds the result of step tb + ostep of chain `ocid` */
        const int n = tb + ostep - 1 - 2 * (nsec - 1);
        if (owner && n >= 0 && n < B) {
            if (FMT != 2 && oc.fir_taps) ring_put(ring_l, ocid, n, flush_word(true), a.ready != nullptr);
            else emit_out(io_l, oc, n, flush_word(false));
        }
    };
    /* The batches in the middle of the block -- all but the first and the last few -- need none of that: every step is busy,
     * every fetched frame and every flushed frame lies inside the block, and from one batch to the next each lane's input
     * pointer, output pointer and ring position advance by a constant.  (tools/cascade_timeline.py: with the checks, the 64-bit
     * address products and the merge of the two step variants, a batch's 16 steps of 55 cycles each took 1370-1500 cycles.) */
    const int bs = ((2 * nsec - 1 + NB - 1) / NB + DEPTH - 1) / DEPTH * DEPTH;       /* first steady batch, a multiple of DEPTH */
    const int be = (B / NB - 1 - DEPTH) / DEPTH * DEPTH;                             /* behind the last: (b + 2 + DEPTH) NB <= B for b < be */
    int b0 = 0;
    for (; b0 < nbatches && (b0 < bs || be <= bs); b0 += DEPTH) {
        batch_any(b0, std::integral_constant<int, 0>{}); batch_any(b0 + 1, std::integral_constant<int, 1>{}); batch_any(b0 + 2, std::integral_constant<int, 2>{});
    }
    if (b0 < be) {
        const unsigned *in_run = inp + (size_t)((b0 + 1 + DEPTH) * NB + (s & (NB - 1))) * io_l.in_stride;
        const size_t in_step = (size_t)NB * io_l.in_stride, out_step = (size_t)NB * io_l.out_stride;
        const int n0 = b0 * NB + ostep - 1 - 2 * (nsec - 1);                         /* the frame this lane flushes in batch b0 */
        const bool to_ring = FMT != 2 && oc.fir_taps != 0;
        const unsigned rmask = (unsigned)ring_l.R - 1u;
        unsigned ridx = ((unsigned)(ring_l.wpos + n0)) & rmask;
        float *rrow = ring_row(ring_l, ocid);
        double *wrow = ring_l.wide ? wide_row(ring_l, ocid) : nullptr;
        unsigned *out_run = io_l.out + (size_t)n0 * io_l.out_stride + (oc.out_io[0] - io_l.out_base);
        const bool more_stores = owner && !to_ring && oc.n_out > 1;
        const bool wt = a.ready != nullptr;
        int n_run = n0;
        auto batch_steady = [&](auto jc) __attribute__((always_inline)) {
            constexpr int j = decltype(jc)::value;
            fast_steps(0);
            ib = hand_from_sample_or_raw<FMT>(rawq[kNext[j]], c_load_mode, c_gain_bits);
            rawq[kNext[j]] = *in_run;
            in_run += in_step;
            if (owner) {
                if (to_ring) {
                    const unsigned w = flush_word(true);
                    if (wt) {                             /* (write-through, like ring_put) */
                        __hip_atomic_store(reinterpret_cast<unsigned *>(rrow) + ridx, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(reinterpret_cast<unsigned *>(rrow) + ridx + (rmask + 1u), w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (wrow) {
                            const unsigned long long wd = (unsigned long long)__double_as_longlong(mulop(__uint_as_float(w)));
                            __hip_atomic_store(reinterpret_cast<unsigned long long *>(wrow) + ((ridx + 3u) & rmask), wd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(reinterpret_cast<unsigned long long *>(wrow) + ((ridx + 3u) & rmask) + (rmask + 1u), wd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    } else {
                        rrow[ridx] = __uint_as_float(w); rrow[ridx + (rmask + 1u)] = __uint_as_float(w);
                        if (wrow) { const double wd = mulop(__uint_as_float(w)); wrow[(ridx + 3u) & rmask] = wd; wrow[((ridx + 3u) & rmask) + (rmask + 1u)] = wd; }
                    }
                } else {
                    const unsigned w = flush_word(false);
                    if (more_stores) emit_out(io_l, oc, n_run, w);      /* (several STOREs of the same value: the general way) */
                    else *out_run = w;
                }
            }
            ridx = (ridx + (unsigned)NB) & rmask;
            out_run += out_step;
            n_run += NB;
        };
        for (; b0 < be; b0 += DEPTH) {
            if (b0 >= 12 && b0 < 36) BQ_STAMP(2 + b0 - 12);
            batch_steady(std::integral_constant<int, 0>{});
            if (b0 + 1 >= 12 && b0 + 1 < 36) BQ_STAMP(2 + b0 + 1 - 12);
            batch_steady(std::integral_constant<int, 1>{});
            if (b0 + 2 >= 12 && b0 + 2 < 36) BQ_STAMP(2 + b0 + 2 - 12);
            batch_steady(std::integral_constant<int, 2>{});
        }
    }
    for (; b0 < nbatches; b0 += DEPTH) {                /* (up to DEPTH - 1 batches past the end: every step masked, nothing stored) */
        batch_any(b0, std::integral_constant<int, 0>{}); batch_any(b0 + 1, std::integral_constant<int, 1>{}); batch_any(b0 + 2, std::integral_constant<int, 2>{});
    }

    /* Inf / NaN.  The loop above computes with IEEE values; the reference's products read exponent 255 as 1.m x 2^128
     * (mulop).  The two only differ when such a value turns up, and then it leaves a trace: an Inf or NaN sample or
     * result makes the section's accumulator non-finite for good (Inf * 0 is NaN), or -- in the block's last frames --
     * sits in its x / y state.  One look at the end of the block, no cost inside the loop: a chain with a trace in any of
     * its sections writes nothing back (the mirror still holds the state the block started from) and its first lane
     * runs the block again in the reference's own order with the reference's products.  Audio never gets here.   */
    BQ_STAMP(28);
    bool replay = false;
    if constexpr (FMT != 2) {
        const unsigned long long ab = (unsigned long long)__double_as_longlong(acc);
        const bool odd = lane_on && ((ab >> 52 & 0x7FF) == 0x7FF || (x1 & 0x7F800000u) == 0x7F800000u || (x2 & 0x7F800000u) == 0x7F800000u ||
                                     (y1 & 0x7F800000u) == 0x7F800000u || (y2 & 0x7F800000u) == 0x7F800000u);
        const unsigned long long m = __ballot(odd);
        const int lane64 = tid & 63, base = lane64 - (lane64 % P);
        const unsigned long long chainmask = (P == 64 ? ~0ull : ((1ull << P) - 1)) << base;
        replay = (m & chainmask) != 0;
    }
    if (lane_on && !replay) {
        int *st = a.buf + sw;
        unsigned long long bits;
        if constexpr (FMT == 2) bits = (unsigned long long)acc;
        else bits = (unsigned long long)__double_as_longlong(acc);
        st[0] = (int)(unsigned)bits; st[1] = (int)(unsigned)(bits >> 32);
        st[2] = (int)x1; st[3] = (int)x2; st[4] = (int)y1; st[5] = (int)y2;
    }
    if constexpr (FMT != 2) {
        if (replay && have_chain && s == 0) {
            /* copies made here, in the branch nobody takes: the callee wants its arguments in memory, and without them the
             * kernel's own arguments would live there for the whole loop */
            const BiquadArgs a2 = a;
            const avdsp_chain c2 = a2.chains[cid];
            cascade_in_reference_order<FMT>(a2, cid, c2);
        }
    }
    if (a.ready) {                                      /* the chains' blocks are in the ring: say so (a chain's lanes sit in one wave) */
        chain_ready_release();
        if (have_chain && s == 0 && c.fir_taps) chain_ready_publish(a.ready, cid, a.seq);
    }
    BQ_STAMP(29);
}

/* ------------------------------------------------------------------------------------------
 * biquad_row<FMT>: the section-pipelined cascade with fewer instructions per step (round 3; formats 4 and 6, cascades of up to
 * 16 sections, one 16-lane row per chain).  Same arithmetic as biquad_pipe, lane = (chain, section), exact for the same reason.
 *
 * What bounds the cascade is one wave's instruction issue: ~4.4 cycles per instruction whatever it is (tools/cascade_lab.hip: DP,
 * SP, dpp, dependent or not), one wave per SIMD, ~1050 dependent steps per block.  biquad_pipe's step is 13 instructions + 1.8 of
 * batch overhead; here it is 10 + 0.6:
 *   - Rows are RIGHT-aligned: a chain's last section sits in lane 15 of its row whatever the section count, section 0 in lane
 *     16 - nsec.  The result of a step then reaches the lane that will store it with ONE dpp (row_newbcast:15 into one bank of four
 *     lanes of one of four registers; three selects per 16 steps sort them out) instead of a rotating register + a select per step.
 *   - The 16 input samples of a batch go global -> LDS -> 16 registers of every lane; the hand-off is v_cndmask_b32_dpp
 *     (previous lane's result, row_shr:1; a section-0 lane takes the staged sample instead, by a lane mask in vcc): no rotating
 *     input register.
 *   - A lane keeps {acc, P, x1, x2, y1, y2 (doubles), hy (its latest result, float bits)} with the first product of its NEXT
 *     compute (P * c0: the operand was fetched a step ago) already added to acc, as biquad_pipe's hand-scheduled step does:
 *         step u:  t = hand-off;  lanes computing frame u-1-2s:  acc += x1 c1 + x2 c2 + y1 c3 + y2 c4 (one fma each, in order);
 *                  hy = (float)acc;  x2 = x1; x1 = P; y2 = y1; y1 = widen(hy);   lanes computing next step:  P = widen(t); acc += P c0
 *     Steady batches run that as four-step asm statements in which the x and y registers rotate roles instead of being moved;
 *     the steps of the block's fill and drain run the same instructions under EXEC masks with the moves spelled out.
 * Measured (tools/cascade_lab.hip, 4096 chains x 16 sections x 1024 frames): 54 cycles per steady step against 95.
 * ---------------------------------------------------------------------------------------- */
/* ACC (format 4, round 5): chains that STORE int samples.  A format-4 STORE takes 31 bits out of the accumulator itself (s31_from_double), not out
 * of the float the next section would get, so the row kernel -- which hands a float on -- was for FIR-feeding chains only and a plain format-4
 * cascade ran on biquad_pipe (4096 chains x 16 sections: 75.5 us against 34.4 in format 6).  Here the step broadcasts the last section's finished
 * ACCUMULATOR instead of its float -- one 64-bit v_mov_b64_dpp row_newbcast in place of the 32-bit one, in front of the next step's first product
 * instead of behind it: the same ten instructions -- and a lane converts the accumulator it ends up holding once per 16 steps (store_word_f4, no
 * branch).  4096 chains x 16 sections: 75.5 -> 46 us; the rest of the way to format 6's 34 is the int sample's two conversions. */
template <int FMT, bool ACC = false>
__global__ __launch_bounds__(kBlock) void biquad_row(const BiquadArgs a)
{
    static_assert(FMT == 4 || FMT == 6, "double-accumulator models");
    static_assert(!ACC || FMT == 4, "the accumulator leaves the cascade in format 4 only");
    flush_f32_subnormals_like_the_reference();
    BQ_STAMP(0);
    __shared__ __attribute__((aligned(16))) unsigned lin[4][2][4][16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, row = lane >> 4, rp = lane & 15;
    const int blk = xcd_remap(blockIdx.x, a.per_xcd);
    const int slot = blk * 16 + wv * 4 + row;
    const Ring ring_l = a.ring;
    const BlockIO io_l = a.io;
    const int B = io_l.nframes;
    /* one load level to the records, a second to coefficients, state and the first samples (biquad_pipe: four) */
    const RowRec rr = a.rows[slot < a.ngroup ? slot : 0];
    /* the cascade's length: the launch's, or (round 5, a launch of rows of SEVERAL lengths: one launch for all cascades of a plan up to 16
     * sections instead of one per length) the wave's -- its four rows have one length, the table is filled up with empty rows */
    const int L = (a.nsec ? a.nsec : __builtin_amdgcn_readfirstlane(rr.pad[0])) - 1;
    const bool have_chain = slot < a.ngroup && rr.cid >= 0;
    const LaneRec lr = a.lanes[(size_t)(have_chain ? slot : 0) * 16 + rp];
    const int sec = rp - (15 - L);                      /* this lane's section (negative: none) */
    const bool lane_on = have_chain && sec >= 0;
    const int cid = rr.cid;
    const int c_load_mode = rr.flags & 0xFF; const unsigned c_gain_bits = rr.gain_bits;
    const bool c_sat = (rr.flags >> 8 & 1) != 0, to_ring = (rr.flags >> 9 & 1) != 0;
    const unsigned long long firstmask = 0x0001000100010001ull << (15 - L);       /* the section-0 lanes */

    /* sample input: lane (row, i) fetches frame 16 b + i of its row's chain, three batches ahead.  Byte offsets from the block's
     * first word, 32 bits (a block is at most 1024 frames x 65536 IOs), clamped to the block's last frame. */
    const char *in_bytes = reinterpret_cast<const char *>(io_l.in);
    const unsigned in_col = (unsigned)(rr.in_io - io_l.in_base) * 4u, in_frame = (unsigned)io_l.in_stride * 4u;
    const unsigned in_max = (unsigned)(B - 1) * in_frame + in_col, in_step = 16u * in_frame;
    unsigned in_off = min((unsigned)rp * in_frame + in_col, in_max);
    auto fetch_next = [&]() __attribute__((always_inline)) -> unsigned {       /* (every lane loads, section or not: see biquad_pipe) */
        const unsigned v = *reinterpret_cast<const unsigned *>(in_bytes + in_off);
        in_off = min(in_off + in_step, in_max);
        return v;
    };
    unsigned r0 = fetch_next(), r1 = fetch_next(), r2 = fetch_next(), r3 = fetch_next();     /* raw words of batches 0, 1, 2, 3 */

    double cd[5] = {0, 0, 0, 0, 0}, acc = 0, dx1 = 0, dx2 = 0, dy1 = 0, dy2 = 0, P = 0;
    unsigned hy = 0;
    int raw_x1 = 0, raw_y1 = 0;                         /* a one-frame block hands them on as x2 / y2 unchanged: as words (mulop drops a zero's sign) */
    if (lane_on) {
        const int *st = a.buf + lr.state_word;
        raw_x1 = st[2]; raw_y1 = st[4];
        const unsigned long long raw = ((unsigned long long)(unsigned)st[1] << 32) | (unsigned)st[0];
        acc = __longlong_as_double((long long)raw);
        for (int k = 0; k < 5; k++) cd[k] = mulop(__int_as_float(a.buf[lr.coef_word + k]));
        dx1 = mulop(__int_as_float(st[2])); dx2 = mulop(__int_as_float(st[3]));
        hy = (unsigned)st[4];
        dy1 = mulop(__int_as_float(st[4])); dy2 = mulop(__int_as_float(st[5]));
    }

    unsigned *mylin = &lin[wv][0][row][0];
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    auto stage = [&](int buf, unsigned raw) __attribute__((always_inline)) { mylin[buf * 64 + rp] = hand_from_sample_or_raw<FMT>(raw, c_load_mode, c_gain_bits).y; };      /* (kLoadRaw: a piece of a longer cascade) */
    auto take = [&](int buf, unsigned (&x)[16]) __attribute__((always_inline)) {
        const u4 *p = reinterpret_cast<const u4 *>(mylin + buf * 64);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const u4 v = p[q];
            x[4 * q] = v[0]; x[4 * q + 1] = v[1]; x[4 * q + 2] = v[2]; x[4 * q + 3] = v[3];
        }
    };
    unsigned xa[16], xb[16];
    stage(0, r0);
    take(0, xa);

    /* results: lane (row, i) stores frame u0 + i - 1 - 2 L of its row's chain after the batch that starts at step u0.  d_j lane 4 q + i
     * = the result of step 4 q + j (row_newbcast of the last section's lane into bank q of d_j). */
    const bool q1 = (rp & 1) != 0, q2 = (rp & 2) != 0;
    auto pick = [&](unsigned d0, unsigned d1, unsigned d2, unsigned d3) __attribute__((always_inline)) -> unsigned {
        const unsigned lo = q1 ? d1 : d0, hi = q1 ? d3 : d2;
        return q2 ? hi : lo;
    };
    /* any batch, any chain: the general way */
    const bool one_store = (rr.flags >> 16 & 0xFF) == 1;
    auto flush = [&](int u0, unsigned w) __attribute__((always_inline)) {
        const int n = u0 + rp - (1 + 2 * L);
        if (have_chain && n >= 0 && n < B) {
            if (to_ring) ring_put(ring_l, cid, n, w, a.ready != nullptr);
            else {
                const unsigned v = !ACC && c_sat ? __float_as_uint(saturate_f32_0db(__uint_as_float(w))) : w;      /* (ACC: w is the stored word already) */
                if (one_store) io_l.out[(size_t)n * io_l.out_stride + (rr.out_io - io_l.out_base)] = v;     /* (the record has the column: no look at the chain) */
                else {
                    const avdsp_chain oc = a.chains[cid];
                    emit_out(io_l, oc, n, v);
                }
            }
        }
    };
    /* steady batches of chains that store once (or feed a FIR ring without the operand copy): a running 32-bit byte offset from a
     * per-lane base -- the ring row, wrapping, or the output column, advancing by 16 frames -- one store, nothing to decide */
    const bool lean = __ballot(have_chain && (rr.flags >> 16 & 0xFF) != 1 && !to_ring) == 0;
    char *obase = to_ring ? reinterpret_cast<char *>(ring_row(ring_l, cid))
                          : reinterpret_cast<char *>(io_l.out + (rr.out_io - io_l.out_base));
    const bool any_ring = __ballot(have_chain && to_ring) != 0;      /* (wave-uniform: the second copy of a ring sample, R floats on) */
    const unsigned omirror = to_ring ? (unsigned)ring_l.R * 4u : 0u;
    /* ... and the operand ring, where the plan keeps one (fir_stream / fir_flow): mulop(sample) as a double at (position + 3) mod R, twice */
    const bool any_wide = any_ring && ring_l.wide != nullptr;
    char *wbase = any_wide && to_ring ? reinterpret_cast<char *>(wide_row(ring_l, cid)) : nullptr;
    const unsigned oinc = to_ring ? 64u : 64u * (unsigned)io_l.out_stride;
    const unsigned owrap = to_ring ? (unsigned)ring_l.R * 4u - 1u : 0xFFFFFFFFu;
    const bool dosat = FMT == 6 && c_sat && !to_ring;
    unsigned ooff = 0;
    const bool wt = a.ready != nullptr;               /* the launch's rings are handed over through ready words: write-through stores (ring_put) */
    auto store_lean = [&](unsigned w, auto wide_c) __attribute__((always_inline)) {
        const unsigned v = dosat ? __float_as_uint(__builtin_amdgcn_fmed3f(__uint_as_float(w), -1.0f, 1.0f)) : w;     /* = saturate_f32_0db for every value but a NaN (the replay's business) */
        if (have_chain) {
            if (wt) {
                __hip_atomic_store(reinterpret_cast<unsigned *>(obase + ooff), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(reinterpret_cast<unsigned *>(obase + ooff + omirror), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                *reinterpret_cast<unsigned *>(obase + ooff) = v;
                if (any_ring) *reinterpret_cast<unsigned *>(obase + ooff + omirror) = v;
            }
            if (decltype(wide_c)::value && wbase) {
                const unsigned woff8 = ((ooff + 12u) & owrap) * 2u;          /* (frame position + 3) mod R, in doubles */
                const double wd = mulop(__uint_as_float(v));
                if (wt) {
                    __hip_atomic_store(reinterpret_cast<unsigned long long *>(wbase + woff8), (unsigned long long)__double_as_longlong(wd), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(reinterpret_cast<unsigned long long *>(wbase + woff8 + 2u * omirror), (unsigned long long)__double_as_longlong(wd), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    *reinterpret_cast<double *>(wbase + woff8) = wd;
                    *reinterpret_cast<double *>(wbase + woff8 + 2u * omirror) = wd;
                }
            }
        }
        ooff = (ooff + oinc) & owrap;
    };

#define AVDSP_ROW_DPP " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
#define AVDSP_ROW_BCAST(D, O, BM) "v_mov_b32_dpp %[" D "], %[" O "] row_newbcast:15 row_mask:0xf bank_mask:" BM "\n\t"
    /* one step, every lane busy: the hand-off, the four remaining products, the new operand, the rounding, the first product of the
     * next step, the widening of the result, its broadcast.  Nothing directly behind an instruction whose result it needs except
     * fma -> fma; every dpp reads a register written at least two instructions earlier.  X2 / YB: the registers whose contents die
     * in this step take the new operand and the new y1. */
#define AVDSP_ROW_STEP(X1, X2, YA, YB, XK, HY, O, D, BM) \
        "v_cndmask_b32_dpp %[t" XK "], %[" HY "], %[" XK "], vcc" AVDSP_ROW_DPP \
        "v_fma_f64 %[an], %[" X1 "], %[c1], %[an]\n\t" \
        "v_fma_f64 %[an], %[" X2 "], %[c2], %[an]\n\t" \
        "v_fma_f64 %[an], %[" YA "], %[c3], %[an]\n\t" \
        "v_fma_f64 %[an], %[" YB "], %[c4], %[an]\n\t" \
        "v_cvt_f64_f32 %[" X2 "], %[t" XK "]\n\t" \
        "v_cvt_f32_f64 %[" O "], %[an]\n\t" \
        "v_fma_f64 %[an], %[" X2 "], %[c0], %[an]\n\t" \
        "v_cvt_f64_f32 %[" YB "], %[" O "]\n\t" \
        AVDSP_ROW_BCAST(D, O, BM)
    /* four steps: (P, x1, x2) enter in (xa, xb, xc) and leave in (xc, xa, xb); y1 / y2 swap twice */
#define AVDSP_ROW_BLOCK4(BM) \
        asm volatile( \
            "s_mov_b64 vcc, %[m0]\n\t" \
            AVDSP_ROW_STEP("xb", "xc", "ya", "yb", "x0", "hy", "o0", "d0", BM) \
            AVDSP_ROW_STEP("xa", "xb", "yb", "ya", "x1", "o0", "o1", "d1", BM) \
            AVDSP_ROW_STEP("xc", "xa", "ya", "yb", "x2", "o1", "o2", "d2", BM) \
            AVDSP_ROW_STEP("xb", "xc", "yb", "ya", "x3", "o2", "o3", "d3", BM) \
            : [an] "+v"(acc), [xa] "+v"(P), [xb] "+v"(dx1), [xc] "+v"(dx2), [ya] "+v"(dy1), [yb] "+v"(dy2), \
              [tx0] "=&v"(t0), [tx1] "=&v"(t1), [tx2] "=&v"(t2), [tx3] "=&v"(t3), \
              [o0] "=&v"(o0), [o1] "=&v"(o1), [o2] "=&v"(o2), [o3] "=&v"(o3), \
              [d0] "+v"(d0), [d1] "+v"(d1), [d2] "+v"(d2), [d3] "+v"(d3) \
            : [hy] "v"(hy), [c0] "v"(cd[0]), [c1] "v"(cd[1]), [c2] "v"(cd[2]), [c3] "v"(cd[3]), [c4] "v"(cd[4]), \
              [x0] "v"(x[k]), [x1] "v"(x[k + 1]), [x2] "v"(x[k + 2]), [x3] "v"(x[k + 3]), [m0] "s"(firstmask) \
            : "vcc")
    /* ACC: what a lane's 64-bit pick becomes -- the ring's float (a FIR follows) or the int sample (SAT0DB, s31_from_double, dither mask) */
    [[maybe_unused]] auto word_of_acc = [&](double d0, double d1, double d2, double d3) __attribute__((always_inline)) -> unsigned {
        const double lo = q1 ? d1 : d0, hi = q1 ? d3 : d2;
        const double X = q2 ? hi : lo;
        const unsigned ringw = __float_as_uint(narrow_f32(X)), outw = store_word_f4(X, c_sat, io_l.store_mask);
        return to_ring ? ringw : outw;
    };
/* (one 64-bit v_mov_b64_dpp row_newbcast; two 32-bit broadcasts of the halves -- the accumulator then pinned in v[2:3] like biquad_row_i64's --
     * measured 2 us slower over cfg3's shape) */
#define AVDSP_ROW_BCAST64(D, BM) "v_mov_b64_dpp %[" D "], %[an] row_newbcast:15 row_mask:0xf bank_mask:" BM "\n\t"
    /* the same step with the finished accumulator broadcast (before the next step's first product goes into it) */
#define AVDSP_ROW_STEP_ACC(X1, X2, YA, YB, XK, HY, O, D, BM) \
        "v_cndmask_b32_dpp %[t" XK "], %[" HY "], %[" XK "], vcc" AVDSP_ROW_DPP \
        "v_fma_f64 %[an], %[" X1 "], %[c1], %[an]\n\t" \
        "v_fma_f64 %[an], %[" X2 "], %[c2], %[an]\n\t" \
        "v_fma_f64 %[an], %[" YA "], %[c3], %[an]\n\t" \
        "v_fma_f64 %[an], %[" YB "], %[c4], %[an]\n\t" \
        "v_cvt_f64_f32 %[" X2 "], %[t" XK "]\n\t" \
        "v_cvt_f32_f64 %[" O "], %[an]\n\t" \
        AVDSP_ROW_BCAST64(D, BM) \
        "v_fma_f64 %[an], %[" X2 "], %[c0], %[an]\n\t" \
        "v_cvt_f64_f32 %[" YB "], %[" O "]\n\t"
#define AVDSP_ROW_BLOCK4_ACC(BM) \
        asm volatile( \
            "s_mov_b64 vcc, %[m0]\n\t" \
            AVDSP_ROW_STEP_ACC("xb", "xc", "ya", "yb", "x0", "hy", "o0", "d0", BM) \
            AVDSP_ROW_STEP_ACC("xa", "xb", "yb", "ya", "x1", "o0", "o1", "d1", BM) \
            AVDSP_ROW_STEP_ACC("xc", "xa", "ya", "yb", "x2", "o1", "o2", "d2", BM) \
            AVDSP_ROW_STEP_ACC("xb", "xc", "yb", "ya", "x3", "o2", "o3", "d3", BM) \
            : [an] "+v"(acc), [xa] "+v"(P), [xb] "+v"(dx1), [xc] "+v"(dx2), [ya] "+v"(dy1), [yb] "+v"(dy2), \
              [tx0] "=&v"(t0), [tx1] "=&v"(t1), [tx2] "=&v"(t2), [tx3] "=&v"(t3), \
              [o0] "=&v"(o0), [o1] "=&v"(o1), [o2] "=&v"(o2), [o3] "=&v"(o3), \
              [d0] "+v"(e0), [d1] "+v"(e1), [d2] "+v"(e2), [d3] "+v"(e3) \
            : [hy] "v"(hy), [c0] "v"(cd[0]), [c1] "v"(cd[1]), [c2] "v"(cd[2]), [c3] "v"(cd[3]), [c4] "v"(cd[4]), \
              [x0] "v"(x[k]), [x1] "v"(x[k + 1]), [x2] "v"(x[k + 2]), [x3] "v"(x[k + 3]), [m0] "s"(firstmask) \
            : "vcc")
    auto steady_steps = [&](const unsigned (&x)[16]) __attribute__((always_inline)) -> unsigned {
        if constexpr (ACC) {
            double e0, e1, e2, e3;
            asm volatile("; e0..e3 start undefined" : "=v"(e0), "=v"(e1), "=v"(e2), "=v"(e3));
#pragma unroll
            for (int k = 0; k < 16; k += 4) {
                unsigned t0, t1, t2, t3, o0, o1, o2, o3;
                if (k == 0) AVDSP_ROW_BLOCK4_ACC("0x1"); else if (k == 4) AVDSP_ROW_BLOCK4_ACC("0x2"); else if (k == 8) AVDSP_ROW_BLOCK4_ACC("0x4"); else AVDSP_ROW_BLOCK4_ACC("0x8");
                { const double t = P; P = dx2; dx2 = dx1; dx1 = t; }
                hy = o3;
            }
            return word_of_acc(e0, e1, e2, e3);
        }
        unsigned d0, d1, d2, d3;                          /* every lane of them is written by one of the four blocks: no initial value */
        asm volatile("; d0..d3 start undefined" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3));
#pragma unroll
        for (int k = 0; k < 16; k += 4) {
            unsigned t0, t1, t2, t3, o0, o1, o2, o3;
            if (k == 0) AVDSP_ROW_BLOCK4("0x1"); else if (k == 4) AVDSP_ROW_BLOCK4("0x2"); else if (k == 8) AVDSP_ROW_BLOCK4("0x4"); else AVDSP_ROW_BLOCK4("0x8");
            { const double t = P; P = dx2; dx2 = dx1; dx1 = t; }
            hy = o3;
        }
        return pick(d0, d1, d2, d3);
    };
    /* one step of the block's fill or drain: the same instructions under EXEC masks, the state moved instead of renamed.
     * cm = the lanes that compute in this step, cn = those that compute in the next one (they take the new operand). */
#define AVDSP_ROW_EDGE(D, BM) \
        asm volatile( \
            "s_mov_b64 %[sv], exec\n\t" \
            "s_mov_b64 vcc, %[m0]\n\t" \
            "v_cndmask_b32_dpp %[t], %[hy], %[xk], vcc" AVDSP_ROW_DPP \
            "s_mov_b64 exec, %[cm]\n\t" \
            "v_fma_f64 %[an], %[x1], %[c1], %[an]\n\t" \
            "v_fma_f64 %[an], %[x2], %[c2], %[an]\n\t" \
            "v_fma_f64 %[an], %[y1], %[c3], %[an]\n\t" \
            "v_fma_f64 %[an], %[y2], %[c4], %[an]\n\t" \
            "v_cvt_f32_f64 %[hy], %[an]\n\t" \
            "v_mov_b64 %[x2], %[x1]\n\t" \
            "v_mov_b64 %[x1], %[p]\n\t" \
            "v_mov_b64 %[y2], %[y1]\n\t" \
            "v_cvt_f64_f32 %[y1], %[hy]\n\t" \
            "s_mov_b64 exec, %[cn]\n\t" \
            "v_cvt_f64_f32 %[p], %[t]\n\t" \
            "s_nop 0\n\t" \
            "v_fma_f64 %[an], %[p], %[c0], %[an]\n\t" \
            "s_mov_b64 exec, %[sv]\n\t" \
            AVDSP_ROW_BCAST(D, "hy", BM) \
            : [an] "+v"(acc), [p] "+v"(P), [x1] "+v"(dx1), [x2] "+v"(dx2), [y1] "+v"(dy1), [y2] "+v"(dy2), [hy] "+v"(hy), \
              [t] "=&v"(t), [sv] "=&s"(sv), [d0] "+v"(d0), [d1] "+v"(d1), [d2] "+v"(d2), [d3] "+v"(d3) \
            : [c0] "v"(cd[0]), [c1] "v"(cd[1]), [c2] "v"(cd[2]), [c3] "v"(cd[3]), [c4] "v"(cd[4]), \
              [xk] "v"(x[k]), [m0] "s"(firstmask), [cm] "s"(cm), [cn] "s"(cn) \
            : "vcc")
    /* section s computes frame u - 1 - 2 s in step u: active while 0 <= u - ustart < B */
    const unsigned ustart = lane_on ? (unsigned)(1 + 2 * sec) : 0x40000000u;
    /* ... with the accumulator: a copy of it taken while the computing lanes are the active ones, broadcast once everybody is */
#define AVDSP_ROW_EDGE_ACC(D, BM) \
        asm volatile( \
            "s_mov_b64 %[sv], exec\n\t" \
            "s_mov_b64 vcc, %[m0]\n\t" \
            "v_cndmask_b32_dpp %[t], %[hy], %[xk], vcc" AVDSP_ROW_DPP \
            "s_mov_b64 exec, %[cm]\n\t" \
            "v_fma_f64 %[an], %[x1], %[c1], %[an]\n\t" \
            "v_fma_f64 %[an], %[x2], %[c2], %[an]\n\t" \
            "v_fma_f64 %[an], %[y1], %[c3], %[an]\n\t" \
            "v_fma_f64 %[an], %[y2], %[c4], %[an]\n\t" \
            "v_cvt_f32_f64 %[hy], %[an]\n\t" \
            "v_mov_b64 %[ac], %[an]\n\t" \
            "v_mov_b64 %[x2], %[x1]\n\t" \
            "v_mov_b64 %[x1], %[p]\n\t" \
            "v_mov_b64 %[y2], %[y1]\n\t" \
            "v_cvt_f64_f32 %[y1], %[hy]\n\t" \
            "s_mov_b64 exec, %[cn]\n\t" \
            "v_cvt_f64_f32 %[p], %[t]\n\t" \
            "s_nop 0\n\t" \
            "v_fma_f64 %[an], %[p], %[c0], %[an]\n\t" \
            "s_mov_b64 exec, %[sv]\n\t" \
            "v_mov_b64_dpp %[" D "], %[ac] row_newbcast:15 row_mask:0xf bank_mask:" BM "\n\t" \
            : [an] "+v"(acc), [p] "+v"(P), [x1] "+v"(dx1), [x2] "+v"(dx2), [y1] "+v"(dy1), [y2] "+v"(dy2), [hy] "+v"(hy), [ac] "+v"(ac), \
              [t] "=&v"(t), [sv] "=&s"(sv), [d0] "+v"(e0), [d1] "+v"(e1), [d2] "+v"(e2), [d3] "+v"(e3) \
            : [c0] "v"(cd[0]), [c1] "v"(cd[1]), [c2] "v"(cd[2]), [c3] "v"(cd[3]), [c4] "v"(cd[4]), \
              [xk] "v"(x[k]), [m0] "s"(firstmask), [cm] "s"(cm), [cn] "s"(cn) \
            : "vcc")
    auto edge_steps = [&](int u0, const unsigned (&x)[16]) __attribute__((always_inline)) -> unsigned {
        if constexpr (ACC) {
            double e0 = 0, e1 = 0, e2 = 0, e3 = 0, ac = 0;
            const unsigned ph = (unsigned)u0 - ustart;
            unsigned long long cn = __ballot(ph < (unsigned)B);
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const unsigned long long cm = cn;
                cn = __ballot(ph + (unsigned)(k + 1) < (unsigned)B);
                unsigned t; unsigned long long sv;
                if ((k & 3) == 0) { if (k == 0) AVDSP_ROW_EDGE_ACC("d0", "0x1"); else if (k == 4) AVDSP_ROW_EDGE_ACC("d0", "0x2"); else if (k == 8) AVDSP_ROW_EDGE_ACC("d0", "0x4"); else AVDSP_ROW_EDGE_ACC("d0", "0x8"); }
                else if ((k & 3) == 1) { if (k == 1) AVDSP_ROW_EDGE_ACC("d1", "0x1"); else if (k == 5) AVDSP_ROW_EDGE_ACC("d1", "0x2"); else if (k == 9) AVDSP_ROW_EDGE_ACC("d1", "0x4"); else AVDSP_ROW_EDGE_ACC("d1", "0x8"); }
                else if ((k & 3) == 2) { if (k == 2) AVDSP_ROW_EDGE_ACC("d2", "0x1"); else if (k == 6) AVDSP_ROW_EDGE_ACC("d2", "0x2"); else if (k == 10) AVDSP_ROW_EDGE_ACC("d2", "0x4"); else AVDSP_ROW_EDGE_ACC("d2", "0x8"); }
                else { if (k == 3) AVDSP_ROW_EDGE_ACC("d3", "0x1"); else if (k == 7) AVDSP_ROW_EDGE_ACC("d3", "0x2"); else if (k == 11) AVDSP_ROW_EDGE_ACC("d3", "0x4"); else AVDSP_ROW_EDGE_ACC("d3", "0x8"); }
            }
            return word_of_acc(e0, e1, e2, e3);
        }
        unsigned d0 = 0, d1 = 0, d2 = 0, d3 = 0;
        const unsigned ph = (unsigned)u0 - ustart;
        unsigned long long cn = __ballot(ph < (unsigned)B);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const unsigned long long cm = cn;
            cn = __ballot(ph + (unsigned)(k + 1) < (unsigned)B);
            unsigned t; unsigned long long sv;
            if ((k & 3) == 0) { if (k == 0) AVDSP_ROW_EDGE("d0", "0x1"); else if (k == 4) AVDSP_ROW_EDGE("d0", "0x2"); else if (k == 8) AVDSP_ROW_EDGE("d0", "0x4"); else AVDSP_ROW_EDGE("d0", "0x8"); }
            else if ((k & 3) == 1) { if (k == 1) AVDSP_ROW_EDGE("d1", "0x1"); else if (k == 5) AVDSP_ROW_EDGE("d1", "0x2"); else if (k == 9) AVDSP_ROW_EDGE("d1", "0x4"); else AVDSP_ROW_EDGE("d1", "0x8"); }
            else if ((k & 3) == 2) { if (k == 2) AVDSP_ROW_EDGE("d2", "0x1"); else if (k == 6) AVDSP_ROW_EDGE("d2", "0x2"); else if (k == 10) AVDSP_ROW_EDGE("d2", "0x4"); else AVDSP_ROW_EDGE("d2", "0x8"); }
            else { if (k == 3) AVDSP_ROW_EDGE("d3", "0x1"); else if (k == 7) AVDSP_ROW_EDGE("d3", "0x2"); else if (k == 11) AVDSP_ROW_EDGE("d3", "0x4"); else AVDSP_ROW_EDGE("d3", "0x8"); }
        }
        return pick(d0, d1, d2, d3);
    };

    const int U = B + 2 * L + 1;                        /* steps u = 0 .. U-1: section s computes frame u - 1 - 2 s */
    const int nb = (U + 15) / 16;
    /* a batch in canonical registers: inputs of batch b in xa, raw words of b+1, b+2, b+3 in r1, r2, r3 */
    auto batch_canon = [&](int b) __attribute__((always_inline)) {
        const int u0 = 16 * b;
        stage((b + 1) & 1, r1);
        const unsigned rn = fetch_next();
        take((b + 1) & 1, xb);
        unsigned w;
        if (u0 >= 2 * L + 1 && u0 + 15 <= B - 1) w = steady_steps(xa);       /* every lane computes in all 16 steps and in the one after */
        else w = edge_steps(u0, xa);
        flush(u0, w);
        r1 = r2; r2 = r3; r3 = rn;
#pragma unroll
        for (int k = 0; k < 16; k++) xa[k] = xb[k];
    };
    /* six steady batches with every queue slot and both register sets under names of their own (a queue that shifts moves
     * registers that loads are still in flight to: see biquad_pipe) */
    auto batch_named = [&](int b, const unsigned (&x)[16], unsigned (&xn)[16], unsigned &rslot, auto lean_c, auto wide_c) __attribute__((always_inline)) {
        if (b >= 12 && b < 36) BQ_STAMP(2 + b - 12);
        stage((b + 1) & 1, rslot);                      /* the raw word of batch b+1 leaves its slot, the fetch of b+4 takes it */
        rslot = fetch_next();
        take((b + 1) & 1, xn);
        const unsigned w = steady_steps(x);
        if constexpr (decltype(lean_c)::value) store_lean(w, wide_c); else flush(16 * b, w);
    };
    auto six = [&](int b, auto lean_c, auto wide_c) __attribute__((always_inline)) {
        batch_named(b, xa, xb, r1, lean_c, wide_c);     batch_named(b + 1, xb, xa, r2, lean_c, wide_c); batch_named(b + 2, xa, xb, r3, lean_c, wide_c);
        batch_named(b + 3, xb, xa, r1, lean_c, wide_c); batch_named(b + 4, xa, xb, r2, lean_c, wide_c); batch_named(b + 5, xb, xa, r3, lean_c, wide_c);
    };
    const int first_steady = (2 * L + 1 + 15) / 16;     /* batches with 16 b >= 2 L + 1 ... */
    const int end_steady = B >= 16 ? (B - 16) / 16 + 1 : 0;      /* ... and 16 b + 15 <= B - 1 */
    /* Round 5 (tools/cascade_timeline.py --block 256): a canonical batch cost ~2 400 cycles whatever its sixteen steps were made of --
     * masked steps of 21 instructions or, tried, steps of 12 -- because it ENDS with `r1 = r2; r2 = r3; r3 = rn`: the last of those
     * moves a register that a load issued at the batch's own start is still in flight to, the compiler puts s_waitcnt vmcnt(0) in
     * front of it, and a batch is shorter than a trip to memory.  So:
     *   - the first three batches (the fill and one steady batch) are written out with named slots like the steady loop's: a fetched
     *     word goes straight into the slot the staged one has left, nothing in flight is moved, and after three the slots hold
     *     batches b + 1, b + 2, b + 3 in order again;
     *   - the batches behind the steady loop fetch only frames the block HAS: the samples of the drain do not exist, and the words of
     *     the few steady batches left over are in the queue already (a block of 96 n + 32 .. 96 n + 80 frames leaves up to two to fetch
     *     the old way). */
    const int nin = (B + 15) / 16;                      /* batches of input the block has */
    auto batch_tail = [&](int bb) __attribute__((always_inline)) {
        const int u0 = 16 * bb;
        stage((bb + 1) & 1, r1);
        unsigned rn = 0;
        if (bb + 4 < nin) rn = fetch_next();
        take((bb + 1) & 1, xb);
        unsigned w;
        if (u0 >= 2 * L + 1 && u0 + 15 <= B - 1) w = steady_steps(xa);
        else w = edge_steps(u0, xa);
        flush(u0, w);
        r1 = r2; r2 = r3; r3 = rn;
#pragma unroll
        for (int k = 0; k < 16; k++) xa[k] = xb[k];
    };
    auto batch_head = [&](int bb, const unsigned (&x)[16], unsigned (&xn)[16], unsigned &rslot) __attribute__((always_inline)) {
        const int u0 = 16 * bb;
        stage((bb + 1) & 1, rslot);
        rslot = fetch_next();
        take((bb + 1) & 1, xn);
        unsigned w;
        if (u0 >= 2 * L + 1 && u0 + 15 <= B - 1) w = steady_steps(x);
        else w = edge_steps(u0, x);
        flush(u0, w);
    };
    int b = 0;
    BQ_STAMP(1);
    if (B >= 96) {                                      /* (three batches and then some: shorter blocks take the plain loop) */
        batch_head(0, xa, xb, r1); batch_head(1, xb, xa, r2); batch_head(2, xa, xb, r3);
#pragma unroll
        for (int k = 0; k < 16; k++) xa[k] = xb[k];
        b = 3;
    } else
        for (; b < nb && b < first_steady; b++) batch_canon(b);
    BQ_STAMP(26);
    if (b + 6 <= end_steady) {
        if (lean) {
            const int n0 = 16 * b + rp - (1 + 2 * L);   /* >= 0 from the first steady batch on */
            ooff = to_ring ? ((unsigned)(ring_l.wpos + n0) & (unsigned)(ring_l.R - 1)) * 4u : (unsigned)n0 * (unsigned)io_l.out_stride * 4u;
            /* (the operand ring's stores are a loop of their own: where no plan keeps one -- the default -- the steady loop is the plain one) */
            if (any_wide) { for (; b + 6 <= end_steady; b += 6) six(b, std::true_type{}, std::true_type{}); }
            else for (; b + 6 <= end_steady; b += 6) six(b, std::true_type{}, std::false_type{});
        } else
            for (; b + 6 <= end_steady; b += 6) six(b, std::false_type{}, std::false_type{});
    }
    BQ_STAMP(27);
    for (; b < nb; b++) batch_tail(b);
    BQ_STAMP(28);
#undef AVDSP_ROW_EDGE_ACC
#undef AVDSP_ROW_BLOCK4_ACC
#undef AVDSP_ROW_STEP_ACC
#undef AVDSP_ROW_BCAST64
#undef AVDSP_ROW_EDGE
#undef AVDSP_ROW_BLOCK4
#undef AVDSP_ROW_STEP
#undef AVDSP_ROW_BCAST
#undef AVDSP_ROW_DPP

    /* Inf / NaN: as in biquad_pipe -- one look at the end of the block, the chain's first lane replays it in the reference's order
     * from the untouched state (the widened x / y values carry an Inf or NaN as a non-finite double) */
    auto nonfinite = [](double d) { return ((unsigned long long)__double_as_longlong(d) >> 52 & 0x7FF) == 0x7FF; };
    const bool odd = lane_on && (nonfinite(acc) || nonfinite(dx1) || nonfinite(dx2) || nonfinite(dy1) || nonfinite(dy2));
    const unsigned long long m = __ballot(odd);
    const bool replay = (m & (0xFFFFull << (16 * row))) != 0;
    if (lane_on && !replay) {
        int *st = a.buf + lr.state_word;
        const unsigned long long bits = (unsigned long long)__double_as_longlong(acc);
        st[0] = (int)(unsigned)bits; st[1] = (int)(unsigned)(bits >> 32);
        st[2] = __float_as_int(narrow_f32(dx1)); st[3] = B == 1 ? raw_x1 : __float_as_int(narrow_f32(dx2));
        st[4] = (int)hy; st[5] = B == 1 ? raw_y1 : __float_as_int(narrow_f32(dy2));
    }
    if (replay && have_chain && sec == 0) {
        const BiquadArgs a2 = a;
        const avdsp_chain c2 = a2.chains[cid];
        cascade_in_reference_order<FMT>(a2, cid, c2);
    }
    if (a.ready) {                                      /* the rows' blocks are in the ring: say so */
        chain_ready_release();
        if (have_chain && rp == 15 && to_ring) chain_ready_publish(a.ready, cid, a.seq);
    }
    BQ_STAMP(29);
}

/* biquad_row_i64: the same arrangement for the int64 model (dsp_biquadSTD.h:34-77).  A lane keeps {acc (with P * c0 of its next compute
 * already added), P, x1, x2, y1 = hy, y2} as 32-bit words; a step is 5 v_mad_i64_i32, the hand-off dpp (straight into the register of
 * the x2 that has just died), the saturation look (hi + 2^27 - 2 against 2^28 - 3, unsigned: one add, one compare, and a branch over
 * the clamp that almost never runs), y = acc >> 28, the word that leaves (acc >> 28 behind SAT0DB, else acc's low word: one
 * v_alignbit with a per-lane shift) and its broadcast: 11 vector + 2 scalar instructions (biquad_pipe<2>: ~20).  The accumulator lives
 * in v[2:3] inside the asm statements (its halves are named there). */
__global__ __launch_bounds__(kBlock) void biquad_row_i64(const BiquadArgs a)
{
    BQ_STAMP(0);
    __shared__ __attribute__((aligned(16))) unsigned lin[4][2][4][16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, row = lane >> 4, rp = lane & 15;
    const int blk = xcd_remap(blockIdx.x, a.per_xcd);
    const int slot = blk * 16 + wv * 4 + row;
    const BlockIO io_l = a.io;
    const int B = io_l.nframes;
    const RowRec rr = a.rows[slot < a.ngroup ? slot : 0];
    const int L = (a.nsec ? a.nsec : __builtin_amdgcn_readfirstlane(rr.pad[0])) - 1;      /* (biquad_row: rows of several lengths in one launch) */
    const bool have_chain = slot < a.ngroup && rr.cid >= 0;
    const LaneRec lr = a.lanes[(size_t)(have_chain ? slot : 0) * 16 + rp];
    const int sec = rp - (15 - L);
    const bool lane_on = have_chain && sec >= 0;
    const int cid = rr.cid;
    const int c_load_mode = rr.flags & 0xFF; const unsigned c_gain_bits = rr.gain_bits;
    const unsigned sh = (rr.flags >> 8 & 1) ? 28u : 0u;          /* SAT0DB: the stored word is acc >> 28 (saturate64_031 of an accumulator the cascade has clamped already), else acc's low word */
    const unsigned omask = (rr.flags >> 10 & 1) ? 0xFFFFFFFFu : (unsigned)io_l.store_mask;      /* (bit 10, with bit 8: kStoreRaw -- a piece hands acc >> 28 on as it is) */
    const unsigned long long firstmask = 0x0001000100010001ull << (15 - L);

    const char *in_bytes = reinterpret_cast<const char *>(io_l.in);
    const unsigned in_col = (unsigned)(rr.in_io - io_l.in_base) * 4u, in_frame = (unsigned)io_l.in_stride * 4u;
    const unsigned in_max = (unsigned)(B - 1) * in_frame + in_col, in_step = 16u * in_frame;
    unsigned in_off = min((unsigned)rp * in_frame + in_col, in_max);
    auto fetch_next = [&]() __attribute__((always_inline)) -> unsigned {
        const unsigned v = *reinterpret_cast<const unsigned *>(in_bytes + in_off);
        in_off = min(in_off + in_step, in_max);
        return v;
    };
    unsigned r0 = fetch_next(), r1 = fetch_next(), r2 = fetch_next(), r3 = fetch_next();

    int ci[5] = {0, 0, 0, 0, 0};
    long long acc = 0;
    unsigned P = 0, x1 = 0, x2 = 0, hy = 0, yp = 0;
    if (lane_on) {
        const int *st = a.buf + lr.state_word;
        acc = (long long)(((unsigned long long)(unsigned)st[1] << 32) | (unsigned)st[0]);
        for (int k = 0; k < 5; k++) ci[k] = a.buf[lr.coef_word + k];
        x1 = (unsigned)st[2]; x2 = (unsigned)st[3]; hy = (unsigned)st[4]; yp = (unsigned)st[5];
    }
    unsigned *mylin = &lin[wv][0][row][0];
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    auto stage = [&](int buf, unsigned raw) __attribute__((always_inline)) { mylin[buf * 64 + rp] = hand_from_sample_or_raw<2>(raw, c_load_mode, c_gain_bits).y; };
    auto take = [&](int buf, unsigned (&x)[16]) __attribute__((always_inline)) {
        const u4 *p = reinterpret_cast<const u4 *>(mylin + buf * 64);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const u4 v = p[q];
            x[4 * q] = v[0]; x[4 * q + 1] = v[1]; x[4 * q + 2] = v[2]; x[4 * q + 3] = v[3];
        }
    };
    unsigned xa[16], xb[16];
    stage(0, r0);
    take(0, xa);

    const bool q1 = (rp & 1) != 0, q2 = (rp & 2) != 0;
    auto pick = [&](unsigned d0, unsigned d1, unsigned d2, unsigned d3) __attribute__((always_inline)) -> unsigned {
        const unsigned lo = q1 ? d1 : d0, hi = q1 ? d3 : d2;
        return (q2 ? hi : lo) & omask;
    };
    auto flush = [&](int u0, unsigned w) __attribute__((always_inline)) {
        const int n = u0 + rp - (1 + 2 * L);
        if (have_chain && n >= 0 && n < B) {
            if ((rr.flags >> 16 & 0xFF) == 1) io_l.out[(size_t)n * io_l.out_stride + (rr.out_io - io_l.out_base)] = w;   /* (the record has the column) */
            else {
                const avdsp_chain oc = a.chains[cid];
                emit_out(io_l, oc, n, w);
            }
        }
    };
    const bool lean = __ballot(have_chain && (rr.flags >> 16 & 0xFF) != 1) == 0;
    char *obase = reinterpret_cast<char *>(io_l.out + (rr.out_io - io_l.out_base));
    const unsigned oinc = 64u * (unsigned)io_l.out_stride;
    unsigned ooff = 0;
    auto store_lean = [&](unsigned w) __attribute__((always_inline)) {
        if (have_chain) *reinterpret_cast<unsigned *>(obase + ooff) = w;
        ooff += oinc;
    };

    const unsigned k_in_range = 0x0FFFFFFDu;          /* hi + (2^27 - 2), unsigned, above this: the accumulator left (-2^59 + 2^33, 2^59) -- checkbiquadsat, dsp_biquadSTD.h:25-32 */
#define AVDSP_ROW_DPP " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
#define AVDSP_ROW_BCAST(D, O, BM) "v_mov_b32_dpp %[" D "], %[" O "] row_newbcast:15 row_mask:0xf bank_mask:" BM "\n\t"
    /* the clamp, for the lanes in SQ: hi >= 2^27 -> 2^59 - 1, hi <= 1 - 2^27 -> -2^59 */
#define AVDSP_ROW_SAT(SQ, LBL) \
        "v_add_u32 %[ta], 0x07fffffe, v3\n\t" \
        "v_cmp_lt_u32 " SQ ", %[kr], %[ta]\n\t" \
        "s_cmp_lg_u64 " SQ ", 0\n\t" \
        "s_cbranch_scc1 .Lbqi_fix" LBL "_%=\n\t" \
        ".Lbqi_ok" LBL "_%=:\n\t"
    /* ... out of line, behind the statement's last instruction: the common path falls through its branch */
#define AVDSP_ROW_FIX(SQ, LBL) \
        ".Lbqi_fix" LBL "_%=:\n\t" \
        "v_ashrrev_i32 %[ta], 31, v3\n\t" \
        "v_xor_b32 %[tb], 0x07ffffff, %[ta]\n\t" \
        "v_not_b32 %[ta], %[ta]\n\t" \
        "v_cndmask_b32 v3, v3, %[tb], " SQ "\n\t" \
        "v_cndmask_b32 v2, v2, %[ta], " SQ "\n\t" \
        "s_branch .Lbqi_ok" LBL "_%=\n\t"
#define AVDSP_ROWI_STEP(X1, X2, Y1, Y2, XK, O, W, D, BM, LBL) \
        "v_mad_i64_i32 v[2:3], %[sj], %[" X1 "], %[c1], v[2:3]\n\t" \
        "v_mad_i64_i32 v[2:3], %[sj], %[" X2 "], %[c2], v[2:3]\n\t" \
        "v_cndmask_b32_dpp %[" X2 "], %[" Y1 "], %[" XK "], vcc" AVDSP_ROW_DPP \
        "v_mad_i64_i32 v[2:3], %[sj], %[" Y1 "], %[c3], v[2:3]\n\t" \
        "v_mad_i64_i32 v[2:3], %[sj], %[" Y2 "], %[c4], v[2:3]\n\t" \
        "v_add_u32 %[ta], 0x07fffffe, v3\n\t" \
        "v_max_u32 %[fl], %[fl], %[ta]\n\t" \
        "v_alignbit_b32 %[" W "], v3, v2, %[sh]\n\t" \
        "v_alignbit_b32 %[" O "], v3, v2, 28\n\t" \
        "v_mad_i64_i32 v[2:3], %[sj], %[" X2 "], %[c0], v[2:3]\n\t" \
        AVDSP_ROW_BCAST(D, W, BM)
    /* four steps: (P, x1, x2) enter in (xa, xb, xc) and leave in (xc, xa, xb); (y1, y2) = the results of the last two steps */
#define AVDSP_ROWI_BLOCK4(BM) \
        asm volatile( \
            "s_mov_b64 vcc, %[m0]\n\t" \
            AVDSP_ROWI_STEP("xb", "xc", "hy", "yp", "x0", "o0", "w0", "d0", BM, "0") \
            AVDSP_ROWI_STEP("xa", "xb", "o0", "hy", "x1", "o1", "w1", "d1", BM, "1") \
            AVDSP_ROWI_STEP("xc", "xa", "o1", "o0", "x2", "o2", "w2", "d2", BM, "2") \
            AVDSP_ROWI_STEP("xb", "xc", "o2", "o1", "x3", "o3", "w3", "d3", BM, "3") \
            : "={v[2:3]}"(acc), [xa] "+v"(P), [xb] "+v"(x1), [xc] "+v"(x2), \
              [o0] "=&v"(o0), [o1] "=&v"(o1), [o2] "=&v"(o2), [o3] "=&v"(o3), \
              [w0] "=&v"(w0), [w1] "=&v"(w1), [w2] "=&v"(w2), [w3] "=&v"(w3), [ta] "=&v"(ta), [fl] "+v"(fl), \
              [d0] "+v"(d0), [d1] "+v"(d1), [d2] "+v"(d2), [d3] "+v"(d3), [sj] "=&s"(sj) \
            : "0"(acc), [hy] "v"(hy), [yp] "v"(yp), [c0] "v"(ci[0]), [c1] "v"(ci[1]), [c2] "v"(ci[2]), [c3] "v"(ci[3]), [c4] "v"(ci[4]), \
              [x0] "v"(x[k]), [x1] "v"(x[k + 1]), [x2] "v"(x[k + 2]), [x3] "v"(x[k + 3]), [m0] "s"(firstmask), [sh] "v"(sh) \
            : "vcc")
    /* 16 steps, every lane busy, WITHOUT the clamp: every step only notes how far its accumulator's high word went (a running
     * maximum of hi + 2^27 - 2, unsigned).  The caller looks at the note after the batch and, if an accumulator left the range,
     * runs the batch again from the state it started with, through the masked steps, which clamp (saturating audio is rare). */
    auto steady_steps = [&](const unsigned (&x)[16], unsigned &fl) __attribute__((always_inline)) -> unsigned {
        unsigned d0, d1, d2, d3;
        asm volatile("; d0..d3 start undefined" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3));
#pragma unroll
        for (int k = 0; k < 16; k += 4) {
            unsigned o0, o1, o2, o3, w0, w1, w2, w3, ta;
            unsigned long long sj;
            if (k == 0) AVDSP_ROWI_BLOCK4("0x1"); else if (k == 4) AVDSP_ROWI_BLOCK4("0x2"); else if (k == 8) AVDSP_ROWI_BLOCK4("0x4"); else AVDSP_ROWI_BLOCK4("0x8");
            { const unsigned t = P; P = x2; x2 = x1; x1 = t; }
            hy = o3; yp = o2;
        }
        return pick(d0, d1, d2, d3);
    };
    /* one step of the fill or drain under EXEC masks (cm: the lanes that compute now, cn: those that compute in the next step) */
#define AVDSP_ROWI_EDGE(D, BM) \
        asm volatile( \
            "s_mov_b64 %[sv], exec\n\t" \
            "s_mov_b64 vcc, %[m0]\n\t" \
            "v_cndmask_b32_dpp %[t], %[hy], %[xk], vcc" AVDSP_ROW_DPP \
            "s_mov_b64 exec, %[cm]\n\t" \
            "v_mad_i64_i32 v[2:3], %[sj], %[x1], %[c1], v[2:3]\n\t" \
            "v_mad_i64_i32 v[2:3], %[sj], %[x2], %[c2], v[2:3]\n\t" \
            "v_mad_i64_i32 v[2:3], %[sj], %[hy], %[c3], v[2:3]\n\t" \
            "v_mad_i64_i32 v[2:3], %[sj], %[yp], %[c4], v[2:3]\n\t" \
            AVDSP_ROW_SAT("%[sq]", "e") \
            "v_mov_b32 %[yp], %[hy]\n\t" \
            "v_alignbit_b32 %[hy], v3, v2, 28\n\t" \
            "v_alignbit_b32 %[w], v3, v2, %[sh]\n\t" \
            "v_mov_b32 %[x2], %[x1]\n\t" \
            "v_mov_b32 %[x1], %[p]\n\t" \
            "s_mov_b64 exec, %[cn]\n\t" \
            "v_mov_b32 %[p], %[t]\n\t" \
            "v_mad_i64_i32 v[2:3], %[sj], %[t], %[c0], v[2:3]\n\t" \
            "s_mov_b64 exec, %[sv]\n\t" \
            AVDSP_ROW_BCAST(D, "w", BM) \
            "s_branch .Lbqi_end_%=\n\t" \
            AVDSP_ROW_FIX("%[sq]", "e") \
            ".Lbqi_end_%=:\n\t" \
            : "={v[2:3]}"(acc), [p] "+v"(P), [x1] "+v"(x1), [x2] "+v"(x2), [hy] "+v"(hy), [yp] "+v"(yp), [w] "+v"(wlast), \
              [t] "=&v"(t), [ta] "=&v"(ta), [tb] "=&v"(tb), [sv] "=&s"(sv), [sq] "=&s"(sq), [sj] "=&s"(sj), \
              [d0] "+v"(d0), [d1] "+v"(d1), [d2] "+v"(d2), [d3] "+v"(d3) \
            : "0"(acc), [c0] "v"(ci[0]), [c1] "v"(ci[1]), [c2] "v"(ci[2]), [c3] "v"(ci[3]), [c4] "v"(ci[4]), \
              [xk] "v"(x[k]), [m0] "s"(firstmask), [cm] "s"(cm), [cn] "s"(cn), [kr] "s"(k_in_range), [sh] "v"(sh) \
            : "vcc", "scc")
    const unsigned ustart = lane_on ? (unsigned)(1 + 2 * sec) : 0x40000000u;
    unsigned wlast = 0;
    auto edge_steps = [&](int u0, const unsigned (&x)[16]) __attribute__((always_inline)) -> unsigned {
        unsigned d0 = 0, d1 = 0, d2 = 0, d3 = 0;
        const unsigned ph = (unsigned)u0 - ustart;
        unsigned long long cn = __ballot(ph < (unsigned)B);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const unsigned long long cm = cn;
            cn = __ballot(ph + (unsigned)(k + 1) < (unsigned)B);
            unsigned t, ta, tb; unsigned long long sv, sq, sj;
            if ((k & 3) == 0) { if (k == 0) AVDSP_ROWI_EDGE("d0", "0x1"); else if (k == 4) AVDSP_ROWI_EDGE("d0", "0x2"); else if (k == 8) AVDSP_ROWI_EDGE("d0", "0x4"); else AVDSP_ROWI_EDGE("d0", "0x8"); }
            else if ((k & 3) == 1) { if (k == 1) AVDSP_ROWI_EDGE("d1", "0x1"); else if (k == 5) AVDSP_ROWI_EDGE("d1", "0x2"); else if (k == 9) AVDSP_ROWI_EDGE("d1", "0x4"); else AVDSP_ROWI_EDGE("d1", "0x8"); }
            else if ((k & 3) == 2) { if (k == 2) AVDSP_ROWI_EDGE("d2", "0x1"); else if (k == 6) AVDSP_ROWI_EDGE("d2", "0x2"); else if (k == 10) AVDSP_ROWI_EDGE("d2", "0x4"); else AVDSP_ROWI_EDGE("d2", "0x8"); }
            else { if (k == 3) AVDSP_ROWI_EDGE("d3", "0x1"); else if (k == 7) AVDSP_ROWI_EDGE("d3", "0x2"); else if (k == 11) AVDSP_ROWI_EDGE("d3", "0x4"); else AVDSP_ROWI_EDGE("d3", "0x8"); }
        }
        return pick(d0, d1, d2, d3);
    };

    auto steady_checked = [&](int u0, const unsigned (&x)[16]) __attribute__((always_inline)) -> unsigned {
        const long long acc0 = acc; const unsigned P0 = P, x10 = x1, x20 = x2, hy0 = hy, yp0 = yp;
        unsigned fl = 0;
        unsigned w = steady_steps(x, fl);
        if (__builtin_expect(__ballot(fl > k_in_range) != 0, 0)) {
            acc = acc0; P = P0; x1 = x10; x2 = x20; hy = hy0; yp = yp0;
            w = edge_steps(u0, x);
        }
        return w;
    };
    const int U = B + 2 * L + 1;
    const int nb = (U + 15) / 16;
    auto batch_canon = [&](int b) __attribute__((always_inline)) {
        const int u0 = 16 * b;
        stage((b + 1) & 1, r1);
        const unsigned rn = fetch_next();
        take((b + 1) & 1, xb);
        unsigned w;
        if (u0 >= 2 * L + 1 && u0 + 15 <= B - 1) w = steady_checked(u0, xa);
        else w = edge_steps(u0, xa);
        flush(u0, w);
        r1 = r2; r2 = r3; r3 = rn;
#pragma unroll
        for (int k = 0; k < 16; k++) xa[k] = xb[k];
    };
    auto batch_named = [&](int b, const unsigned (&x)[16], unsigned (&xn)[16], unsigned &rslot, auto lean_c) __attribute__((always_inline)) {
        if (b >= 12 && b < 36) BQ_STAMP(2 + b - 12);
        stage((b + 1) & 1, rslot);
        rslot = fetch_next();
        take((b + 1) & 1, xn);
        const unsigned w = steady_checked(16 * b, x);
        if constexpr (decltype(lean_c)::value) store_lean(w); else flush(16 * b, w);
    };
    auto six = [&](int b, auto lean_c) __attribute__((always_inline)) {
        batch_named(b, xa, xb, r1, lean_c);     batch_named(b + 1, xb, xa, r2, lean_c); batch_named(b + 2, xa, xb, r3, lean_c);
        batch_named(b + 3, xb, xa, r1, lean_c); batch_named(b + 4, xa, xb, r2, lean_c); batch_named(b + 5, xb, xa, r3, lean_c);
    };
    const int first_steady = (2 * L + 1 + 15) / 16;
    const int end_steady = B >= 16 ? (B - 16) / 16 + 1 : 0;
    /* (round 5, as in biquad_row: the first three batches with named slots, the batches behind the steady loop without fetches of
     * frames the block does not have -- a canonical batch ends by moving a register its own fetch is still in flight to) */
    const int nin = (B + 15) / 16;
    auto batch_tail = [&](int bb) __attribute__((always_inline)) {
        const int u0 = 16 * bb;
        stage((bb + 1) & 1, r1);
        unsigned rn = 0;
        if (bb + 4 < nin) rn = fetch_next();
        take((bb + 1) & 1, xb);
        unsigned w;
        if (u0 >= 2 * L + 1 && u0 + 15 <= B - 1) w = steady_checked(u0, xa);
        else w = edge_steps(u0, xa);
        flush(u0, w);
        r1 = r2; r2 = r3; r3 = rn;
#pragma unroll
        for (int k = 0; k < 16; k++) xa[k] = xb[k];
    };
    auto batch_head = [&](int bb, const unsigned (&x)[16], unsigned (&xn)[16], unsigned &rslot) __attribute__((always_inline)) {
        const int u0 = 16 * bb;
        stage((bb + 1) & 1, rslot);
        rslot = fetch_next();
        take((bb + 1) & 1, xn);
        unsigned w;
        if (u0 >= 2 * L + 1 && u0 + 15 <= B - 1) w = steady_checked(u0, x);
        else w = edge_steps(u0, x);
        flush(u0, w);
    };
    int b = 0;
    BQ_STAMP(1);
    if (B >= 96) {
        batch_head(0, xa, xb, r1); batch_head(1, xb, xa, r2); batch_head(2, xa, xb, r3);
#pragma unroll
        for (int k = 0; k < 16; k++) xa[k] = xb[k];
        b = 3;
    } else
        for (; b < nb && b < first_steady; b++) batch_canon(b);
    BQ_STAMP(26);
    if (b + 6 <= end_steady) {
        if (lean) {
            const int n0 = 16 * b + rp - (1 + 2 * L);
            ooff = (unsigned)n0 * (unsigned)io_l.out_stride * 4u;
            for (; b + 6 <= end_steady; b += 6) six(b, std::true_type{});
        } else
            for (; b + 6 <= end_steady; b += 6) six(b, std::false_type{});
    }
    BQ_STAMP(27);
    for (; b < nb; b++) batch_tail(b);
    BQ_STAMP(28);
#undef AVDSP_ROWI_EDGE
#undef AVDSP_ROWI_BLOCK4
#undef AVDSP_ROWI_STEP
#undef AVDSP_ROW_SAT
#undef AVDSP_ROW_FIX
#undef AVDSP_ROW_BCAST
#undef AVDSP_ROW_DPP
    if (lane_on) {
        int *st = a.buf + lr.state_word;
        st[0] = (int)(unsigned)(unsigned long long)acc; st[1] = (int)(unsigned)((unsigned long long)acc >> 32);
        st[2] = (int)x1; st[3] = (int)x2; st[4] = (int)hy; st[5] = (int)yp;
    }
    BQ_STAMP(29);
}

/* lane per chain, the reference's loop order: the cross-check path ("biquad_impl" 0) */
template <int FMT>
__global__ __launch_bounds__(64) void biquad_simple(const BiquadArgs a)
{
    if constexpr (FMT != 2) flush_f32_subnormals_like_the_reference();
    const int slot = blockIdx.x * 64 + threadIdx.x;
    const int cid = slot < a.ngroup ? a.group[slot] : -1;
    if (cid >= 0) cascade_in_reference_order<FMT>(a, cid, a.chains[cid]);
    if (a.ready) {
        chain_ready_release();
        if (cid >= 0 && a.chains[cid].fir_taps) chain_ready_publish(a.ready, cid, a.seq);
    }
}

/* ------------------------------------------------------------------------------------------
 * FIR
 * ---------------------------------------------------------------------------------------- */
struct FirArgs {
    int            *buf;
    const avdsp_chain *chains;
    const int      *group;
    int             ngroup;
    Ring            ring;
    int             per_xcd;
    int             gpc;             /* groups of 16 tap positions per LDS chunk */
    int             hs_cap;          /* doubles reserved for the taps image */
    int             win_row;         /* doubles per row of the transposed window image */
    BlockIO         io;
};

/* groups of 16 tap positions per operand set: NG = 1 (4 MFMAs) keeps the kernel at 96 VGPRs = 5 waves per
 * SIMD, which beats deeper prefetch when the grid fills the GPU (A/B: -3 %); with at most two workgroups per
 * CU there is nothing to interleave and sets of 2 hide more LDS latency.  The images carry the margin of the
 * deeper variant.                                                                                           */
constexpr int kNG  = 2;
constexpr int kMaxGpc = 56;          /* 896 tap positions per chunk: <= 27 KB of LDS, 5 workgroups per CU, no spills */
constexpr int kFirPad = 1024;        /* frames per launch the window image is laid out for (4 tiles) */

/* the operand prefetch runs up to kNG + 1 groups past the end of a chunk: both images carry that margin */
__host__ __device__ inline int fir_hs_len(int gpc) { return 16 * gpc + 16 * (kNG + 1) + 32; }
__host__ __device__ inline int fir_win_off(int gpc) { return 16 * gpc + 16 * (kNG + 1) + 8; }
__host__ __device__ inline int fir_win_len(int gpc) { return kFirPad + 16 * gpc + 16 * (kNG + 2); }
__host__ __device__ inline int fir_win_row(int gpc)
{
    int r = (fir_win_len(gpc) + 15) >> 4;
    return r + ((16 - (r & 31)) & 31);               /* next value that is 16 mod 32 */
}
__device__ __forceinline__ int win_pos(int w, int row) { return (w & 15) * row + (w >> 4); }
constexpr int kHRegs = (16 * kMaxGpc + 16 * (kNG + 1) + 32 + kBlock - 1) / kBlock;              /* 6 */
constexpr int kWRegs = (kFirPad + 16 * kMaxGpc + 16 * (kNG + 2) + kBlock - 1) / kBlock;         /* 10: slots enumerate w in blocks of 256 */

/* FIR-only chains: the FIR's input is (float)X of the load stage; append it to the ring first */
template <int FMT>
__device__ __forceinline__ void fir_append_input(const FirArgs &a, const avdsp_chain &c, int cid)
{
    const int B = a.io.nframes;
    const unsigned *inp = a.io.in + (c.in_io - a.io.in_base);
    for (int q = threadIdx.x; q < B; q += blockDim.x) {
        unsigned raw = inp[(size_t)q * a.io.in_stride];
        if constexpr (FMT == 6) raw = ftz_bits(raw);           /* fir_mfma runs in the default MODE: flush by hand */
        ring_put(a.ring, cid, q, ftz_bits(narrow_stage<FMT>(load_stage<FMT>(raw, c.load_mode, c.gain_bits))));
    }
    __syncthreads();
}

/* One chunk = tap positions m in [mlo, mlo + 16*gpc) plus the prefetch overrun.  Its operands travel
 * global -> registers (issued one chunk ahead, under the previous chunk's MFMAs) -> LDS as DOUBLES:
 * converted, and flushed to +0 when the float is subnormal (dspMulFloatDouble, dsp_ieee754.h:383-386),
 * once per chunk instead of once per MFMA.
 *   taps    hs[u] = h[mlo + u]                                  (0 outside [0, T))
 *   window  sample q sits at w = q + mlo + fir_win_off(gpc), stored TRANSPOSED in 16 rows,
 *           pos(w) = (w & 15) * row + (w >> 4) with row = 16 (mod 32): the MFMA B operand
 *           x[16(a0+j) - m - k] is 16 consecutive doubles over j, and the two k sharing a
 *           ds_read_b64 pass use disjoint bank halves.  q >= B (the future) reads as 0.          */
struct ChunkRegs { float h[kHRegs], x[kWRegs]; };

/* Which window position thread `tid` handles in its u-th slot.  The image is a transpose (reads walk w
 * in steps of 16), so the 16 lanes of an LDS write group take 16 positions of ONE row -- consecutive
 * doubles, conflict free -- at the price of a 64-byte stride between their global loads (the four
 * groups of a wave take adjacent rows, i.e. adjacent floats of the same cache lines).               */
__device__ __forceinline__ int fir_win_slot(int u, int tid)
{
    const int idx = u * kBlock + tid, i = idx & 15, t = idx >> 4;
    return 16 * (16 * (t >> 4) + i) + (t & 15);
}

__device__ __forceinline__ void fir_chunk_fetch(const FirArgs &a, const avdsp_chain &c, int cid, int mlo, ChunkRegs &r)
{
    const int T = c.fir_taps, B = a.io.nframes, tid = threadIdx.x;
    const float *taps = reinterpret_cast<const float *>(a.buf + c.fir_coef_word);
    const float *ringrow = ring_row(a.ring, cid);
    const int rmask = a.ring.R - 1, wlen = fir_win_len(a.gpc), woff = fir_win_off(a.gpc);
#pragma unroll
    for (int u = 0; u < kHRegs; u++) { const int t = mlo + u * kBlock + tid; r.h[u] = (t >= 0 && t < T) ? taps[t] : 0.0f; }
#pragma unroll
    for (int u = 0; u < kWRegs; u++) {
        const int w = fir_win_slot(u, tid), q = w - mlo - woff;
        r.x[u] = (w < wlen && q < B) ? ringrow[(a.ring.wpos + q) & rmask] : 0.0f;
    }
}

__device__ __forceinline__ void fir_chunk_to_lds(const FirArgs &a, const ChunkRegs &r, double *hs, double *xs, int row)
{
    const int tid = threadIdx.x, hlen = fir_hs_len(a.gpc), wlen = fir_win_len(a.gpc);
#pragma unroll
    for (int u = 0; u < kHRegs; u++) { const int i = u * kBlock + tid; if (i < hlen) hs[i] = mulop(r.h[u]); }
#pragma unroll
    for (int u = 0; u < kWRegs; u++) { const int w = fir_win_slot(u, tid); if (w < wlen) xs[win_pos(w, row)] = mulop(r.x[u]); }
}

typedef double v4f64 __attribute__((ext_vector_type(4)));
typedef double v2f64 __attribute__((ext_vector_type(2)));

/* FIR as a dense contraction on v_mfma_f64_16x16x4_f64.
 *   Y[i][a] = y[16a + i] = sum_m A[i][m] * Bm[m][a],   A[i][m] = h[m + i],   Bm[m][a] = x[16a - m],
 *   m = -15 .. T-1, four values of m per MFMA, ascending = the reference's tap order (dsp_firSTD.h:45-50);
 *   products of two floats are exact in f64, so each accumulator is the reference's sequential sum.
 * One workgroup (4 waves) per channel, one wave per column tile of 16 (256 frames).  The tap range is
 * walked in LDS chunks while the accumulators stay in registers; the small chunk images let five
 * workgroups share a CU, so one workgroup's staging hides under the others' MFMAs.               */
template <int FMT, int NG>
__global__ __launch_bounds__(kBlock, NG == 1 ? 5 : 4) void fir_mfma(const FirArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int slot = xcd_remap(blockIdx.x, a.per_xcd);
    if (slot >= a.ngroup) return;
    const int cid = a.group[slot];
    const avdsp_chain c = a.chains[cid];
    double *hs = lds, *xs = lds + a.hs_cap;
    const int row = a.win_row, gpc = a.gpc;
    const int T = c.fir_taps, B = a.io.nframes;
    if (c.nsec == 0) fir_append_input<FMT>(a, c, cid);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ij = lane & 15, k = lane >> 4;
    const bool busy = wave * kTileFrames < B;           /* waves beyond the block's last tile only help staging */
    const double *hp = hs + k + ij;                     /* h[mlo + 16 g + 4 s + k + i] at [16 g + 4 s] */
    const double *xp[4];                                /* x[16(a0+j) - (mlo + 16 g + 4 s + k)] at [-g] */
    const int f0 = 16 * 16 * wave + fir_win_off(gpc) - k;
#pragma unroll
    for (int s = 0; s < 4; s++) xp[s] = xs + win_pos(f0 - 4 * s, row) + ij;
    v4f64 acc = {0.0, 0.0, 0.0, 0.0};

    double ha[4 * NG], xa[4 * NG], hb[4 * NG], xb[4 * NG];
    auto load_set = [&](double *h, double *x, int g) {
#pragma unroll
        for (int q = 0; q < NG; q++)
#pragma unroll
            for (int s = 0; s < 4; s++) { h[4 * q + s] = hp[16 * (g + q) + 4 * s]; x[4 * q + s] = xp[s][-(g + q)]; }
    };
    auto mfma_set = [&](const double *h, const double *x) {
#pragma unroll
        for (int u = 0; u < 4 * NG; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(h[u], x[u], acc, 0, 0, 0);
    };

    const int total_groups = (T + 15 + 15) >> 4;        /* tap positions m = -15 .. T-1 in groups of 16 */
    /* Workgroups that share a CU were dispatched together and do identical work: left alone they hit
     * their chunk boundaries (barriers + LDS refill, no MFMA) in lockstep.  A first chunk of 1/4 .. 4/4
     * of the normal size, chosen by dispatch round, keeps their refills apart.                     */
    const int phase = (blockIdx.x >> 8) & 3;
    int glen = max(NG, (gpc * (phase + 1) / 4) / NG * NG);
    ChunkRegs regs;
    fir_chunk_fetch(a, c, cid, -15, regs);
    for (int g0 = 0; g0 < total_groups; g0 += glen, glen = gpc) {
        __syncthreads();                                /* everyone is done reading the previous chunk */
        fir_chunk_to_lds(a, regs, hs, xs, row);
        __syncthreads();
        if (g0 + glen < total_groups) fir_chunk_fetch(a, c, cid, -15 + 16 * (g0 + glen), regs);
        if (busy) {
            const int ng = min(glen, total_groups - g0);
            const int nsets = (ng + NG - 1) / NG;
            load_set(ha, xa, 0);
            int st = 0;
            for (; st + 2 <= nsets; st += 2) {
                load_set(hb, xb, NG * (st + 1));
                __builtin_amdgcn_sched_barrier(0);
                mfma_set(ha, xa);
                __builtin_amdgcn_sched_barrier(0);
                load_set(ha, xa, NG * (st + 2));
                __builtin_amdgcn_sched_barrier(0);
                mfma_set(hb, xb);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (st < nsets) mfma_set(ha, xa);
        }
    }
    /* C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg */
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int n = 16 * (16 * wave + ij) + k + 4 * r;
        if (n < B) {
            unsigned word = store_stage<FMT>(acc[r], c.sat, a.io.store_mask);
            if constexpr (FMT == 6) word = ftz_bits(word);      /* default MODE here: flush the float by hand */
            emit_out(a.io, c, n, word);
        }
    }
}


/* ------------------------------------------------------------------------------------------
 * fir_tile<FMT, R>: the FIR contraction with one WAVE per (channel, tile of 256 R frames) and nothing shared
 * between waves -- no workgroup barrier anywhere in the tap loop.
 *
 *   y[F0 + 16R a + 16r + i] = sum_m A_r[i][m] * Bm[m][a],   A_r[i][m] = h[m + 16r + i],   Bm[m][a] = x[F0 + 16R a - m]
 *   r = 0 .. R-1 (row tiles, one accumulator each), i = 0 .. 15, a = 0 .. 15, m = -16R .. T-1 in k-steps of 4, ascending
 *   = the reference's tap order (dsp_firSTD.h:45-50): every accumulator is the reference's sequential sum, bit for bit.
 *
 * The row tiles share the window operand Bm, and A_r at step m is A_0 at step m + 16 r: each taps operand that is read
 * from LDS serves R MFMAs, 4 k-steps apart, out of a register queue.  One ds_read_b64 of taps and one of the window
 * per k-step feed R MFMAs (fir_mfma reads two per MFMA), and every address inside an unrolled group of 16 k-steps is
 * the lane's base plus an immediate.
 *
 * Operands: the taps are converted to double ONCE, when the plan is made (avdsp_hip_prog_add_plan: Hbuf[j] =
 * mulop(h[j - 64]), zero padded on both sides), so staging them is a copy.  The window comes from the chain's float ring
 * and is converted while it is staged, 15 R*16 + 4 CK frames per chunk of CK k-steps, into a transposed image
 * pos(u) = (u mod 16R) * row + u / 16R: the 16 lanes of one k read 16 consecutive doubles.
 * ---------------------------------------------------------------------------------------- */
constexpr int kTapsLead = 64;                /* zeros in front of a chain's taps in the f64 copy */
constexpr int kTapsTail = 384;               /* zeros behind them (the last k-steps and the operand prefetch read on) */
__host__ __device__ inline int taps64_pitch(int max_taps) { return (kTapsLead + max_taps + kTapsTail + 1) & ~1; }

/* BIG (R = 1 only): chunks twice as long for launches that leave a SIMD one wave at most -- nothing hides a chunk boundary there
 * (2200 cycles each, 16 % of a wave's life on 256 chains x 4096 taps: tools/fir_timeline.py), and a workgroup may have the CU's LDS to
 * itself (35 KB per wave) */
template <int R, bool BIG = false> struct TileGeom {
    static_assert(!BIG || R == 1, "long chunks: one row tile");
    static constexpr int NR = 16 * R;                        /* rows of the output tile: frames per column step */
    static constexpr int FW = 256 * R;                       /* frames per wave */
    static constexpr int WPC = 4 / R;                        /* waves per channel and launch */
    static constexpr int QD = 4 * (R - 1);                   /* k-steps a taps operand waits for its last use */
    static constexpr int CKMAX = R == 4 ? 96 : R == 2 ? 128 : BIG ? 320 : 160;       /* k-steps per chunk, multiple of 16 */
    static constexpr int ROW = R == 4 ? 23 : R == 2 ? 33 : BIG ? 97 : 57;            /* doubles per window row, odd: >= 16 + ceil(4 (CKMAX-1) / NR) */
    static constexpr int HNEED = 4 * CKMAX + 16 * (R - 1) + 28;              /* doubles of a chunk's taps image */
    static constexpr int HLEN = (HNEED + 127) / 128 * 128;                   /* ... in whole 1-KiB pieces of the LDS-DMA that fills it */
    static constexpr int WLEN = NR * ROW + (NR * ROW & 1);
    static constexpr int LDS_DOUBLES = 2 * HLEN + WLEN;      /* per wave: two taps images (one being filled), one window image */
    static constexpr int NW = (NR * ROW + 63) / 64;          /* window elements a lane stages per chunk */
};
static_assert(TileGeom<4>::ROW >= 16 + (4 * (TileGeom<4>::CKMAX - 1) + 63) / 64 && TileGeom<2>::ROW >= 16 + (4 * (TileGeom<2>::CKMAX - 1) + 31) / 32 &&
              TileGeom<1>::ROW >= 16 + (4 * (TileGeom<1>::CKMAX - 1) + 15) / 16 &&
              TileGeom<1, true>::ROW >= 16 + (4 * (TileGeom<1, true>::CKMAX - 1) + 15) / 16, "window rows hold a chunk");

struct FirTileArgs {
    int *buf; const avdsp_chain *chains; const int *group; int ngroup; Ring ring; int per_xcd;
    const double *taps64; int pitch64;       /* f64 copy of the taps, [chain id][pitch64] */
    BlockIO io;
    unsigned *ready; unsigned seq; unsigned *timeouts;      /* chain_ready_wait: null = the launch is ordered behind its cascades by the stream / an event */
    int ready_acquire;                                      /* 0: the words were set by a kernel behind the cascade (no acquire needed, chain_ready_wait) */
    int wpc_shift;                           /* fir_tile: log2 of the tiles (waves) a chain has in THIS launch: ceil(frames / tile), 1, 2 or 4, at most
                                                TileGeom::WPC.  Round 5: a block of 256 frames is ONE tile of a one-row-tile wave, and with the waves of a
                                                workgroup fixed at the four tiles of a 1024-frame block three of them left at once while the workgroup kept its
                                                LDS -- two live waves per CU: 4096 chains x 256 frames took 289 us, eight rounds of 512 waves */
#ifdef AVDSP_FIR_STAMPS
    unsigned long long *stamps;              /* diagnostic build (tools/fir_timeline.py): 32 s_memtime stamps per wave */
#endif
};
#ifdef AVDSP_FIR_STAMPS
#define FIR_STAMP(i) do { if (lane == 0 && (i) <= 30) a.stamps[(size_t)(blockIdx.x * 4 + stamp_row) * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define FIR_STAMP_CHUNK() do { if (stamp_i <= 21) FIR_STAMP(stamp_i); stamp_i++; } while (0)   /* three per chunk, the first seven chunks */
#define FIR_REALTIME(i) do { if (lane == 0) a.stamps[(size_t)(blockIdx.x * 4 + stamp_row) * 32 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FIR_STAMP(i) do { } while (0)
#define FIR_STAMP_CHUNK() do { } while (0)
#define FIR_REALTIME(i) do { } while (0)
#endif

/* one taps double per lane and k-step, the window double of lane (a, k) likewise; offsets inside a group of 16 k-steps */
template <int R, bool BIG = false> __device__ __forceinline__ constexpr int win_off(int j)
{
    constexpr int NR = 16 * R;
    /* row part ((-4 j) mod NR) * ROW, entry part -(ceil(4 j / NR)), shifted so that the smallest offset of a group is 0 */
    return ((NR - (4 * j) % NR) % NR) * TileGeom<R, BIG>::ROW - (4 * j + NR - 1) / NR + (64 / NR - 1);
}

/* SPLIT (R = 1, opt-in "fir_split"): a tile's k-steps cut in two, one wave each, the two partial sums added at the end -- NOT the
 * reference's summation order any more (results within north_star's 1e-6, not its bits), which is why it is an option: a launch
 * of at most one tile per SIMD (cfg4: 256 chains x 4 tiles) then has two waves per SIMD, whose MFMAs interleave and whose chunk
 * boundaries hide under each other.  The two halves of a tile are neighbouring waves of one workgroup and meet through LDS. */
/* LEAN (round 4): a chunk boundary with a third of the vector instructions -- one masked window offset per lane and chunk (the ring
 * holds every sample twice, so a window never wraps), no look at every sample for Inf / NaN / subnormals (subnormals: the wave's
 * MODE flushes them in the conversion; Inf / NaN: seen in the tile's sums at the end, such a tile is summed again the reference's
 * way), the taps copy in whole pieces.  While one wave of a SIMD streams MFMAs the other's boundary gets a vector instruction in
 * every few hundred cycles (they share the FP64 datapath): 140 instructions outlast the partner's chunk and what is left of them
 * when it ends is idle matrix pipe; 50 do not.  tools/fir_boundary_lab.sh (both forms on one box): 4096 chains 0.4929 -> 0.4857 ms per
 * launch with the chip to itself, 256 chains x 4096 taps 41.6 -> 40.3 us.  NOT where the next blocks' cascades run beside a single
 * round of FIR waves (a shard of the program): those cascades live on exactly the bubbles this removes, and the lean boundary in
 * turn no longer fits under the partner's chunk once a cascade wave takes its share of the slots -- a 512-chain shard's FIR alone
 * 68.4 -> 66.7 us, its step 87.6 -> 96.6; 2048 chains 247.2 -> 244.6 us alone, the step 0.259 -> 0.269 ms -- so such launches keep
 * the long boundary (launch_fir chooses). */
template <int FMT, int R, bool BIG = false, bool SPLIT = false, bool LEAN = true>
__global__ __launch_bounds__(kBlock, 2) void fir_tile(const FirTileArgs a)
{
    static_assert(!SPLIT || (R == 1 && !BIG), "the tap split: one row tile, ordinary chunks");
    using G = TileGeom<R, BIG>;
    constexpr int NR = G::NR;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    if constexpr (LEAN) flush_f32_subnormals_like_the_reference();          /* (win_store's conversions; the stores below still flush by hand) */
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int B = a.io.nframes;
    const int blk = xcd_remap(blockIdx.x, a.per_xcd);
    /* the tiles a chain has in this launch: 1 << wsh of the G::WPC a 1024-frame block needs (the workgroup: 4 >> wsh chains) */
    const int wsh = SPLIT ? (G::WPC == 4 ? 2 : G::WPC == 2 ? 1 : 0) : a.wpc_shift;
    /* FIR-only chains: the FIR's input is (float)X of the load stage; the workgroup's chains get theirs appended first
     * (the waves of one chain sit in one workgroup) */
    {
        FirArgs fa{};
        fa.ring = a.ring; fa.io = a.io;
        for (int q = 0; q < (SPLIT ? 1 : 4 >> wsh); q++) {
            const int sl = SPLIT ? blk >> 1 : blk * (4 >> wsh) + q;      /* (SPLIT: the workgroup is two tiles of one chain) */
            if (sl < a.ngroup) {
                const int ci = a.group[sl];
                const avdsp_chain cc = a.chains[ci];
                if (cc.nsec == 0) fir_append_input<FMT>(fa, cc, ci);
            }
        }
        __syncthreads();
    }
    const int unit = SPLIT ? (blk * 4 + wv) >> 1 : blk * 4 + wv;
    [[maybe_unused]] const int half = (blk * 4 + wv) & 1;
    const int slot = unit >> wsh, F0 = (unit & ((1 << wsh) - 1)) * G::FW;
    /* R = 4: the workgroup's four waves are four chains over the same frames, and their tiles leave together (one barrier, at
     * the very end); a wave without a unit only attends that barrier */
    [[maybe_unused]] int *xchg = reinterpret_cast<int *>(lds + (size_t)4 * G::LDS_DOUBLES);      /* [wave]: out_io of its chain, or -1 */
    if (slot >= a.ngroup || F0 >= B) {
        if constexpr (R >= 2) {
            if (lane == 0) xchg[wv] = -1;
            __syncthreads();
        }
        if constexpr (SPLIT) __syncthreads();
        return;
    }
    /* from here on a wave is on its own: no barrier in the tap loop */
    [[maybe_unused]] const int stamp_row = wv;
    FIR_STAMP(0); FIR_REALTIME(29);
#ifdef AVDSP_FIR_STAMPS
    if (lane == 0) a.stamps[(size_t)(blockIdx.x * 4 + wv) * 32 + 31] = ((unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) << 32) |
                                                                       __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));    /* HW_ID | XCC_ID */
#endif
    /* the wave's unit -- chain, tap count, ring row, taps -- is the same for all its lanes: say so (scalar registers, scalar
     * address arithmetic, and loads in the base + 32-bit offset form) */
    const int cid = __builtin_amdgcn_readfirstlane(a.group[slot]);
    const avdsp_chain c = a.chains[cid];
    const int T = __builtin_amdgcn_readfirstlane(c.fir_taps);
    if (a.ready && c.nsec) chain_ready_wait(a.ready, cid, a.seq, a.timeouts, a.ready_acquire != 0);       /* the chain's cascade of this launch has left its block in the ring */
    double *hs = lds + (size_t)wv * G::LDS_DOUBLES, *ws = hs + 2 * G::HLEN;     /* taps images at hs and hs + HLEN */
    const double *hbuf = a.taps64 + (size_t)cid * a.pitch64;
    const float *ringrow = ring_row(a.ring, cid);
    const int rmask = a.ring.R - 1;

    const int i16 = lane & 15, k = lane >> 4;
    /* k-steps: m_s = -NR + 4 s, s = 0 .. S-1, S a multiple of 16; chunks of ck <= CKMAX of them */
    const int Sall = (((T + NR + 3) >> 2) + 15) & ~15;
    /* this wave's k-steps [Sb, S): all of them, or (SPLIT) the first or the second half, cut at a multiple of 16 */
    const int Sb = SPLIT && half ? ((Sall >> 1) + 15) & ~15 : 0;
    const int S = SPLIT && !half ? ((Sall >> 1) + 15) & ~15 : Sall;
    const int nch = max(1, (S - Sb + G::CKMAX - 1) / G::CKMAX);
    const int ck = max(16, (((S - Sb + nch - 1) / nch) + 15) & ~15);

    v4f64 acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = v4f64{0.0, 0.0, 0.0, 0.0};

    /* window element e = lane + 64 t of a chunk's image: row e % NR, entry e / NR */
    const int wrow = lane % NR, went = lane / NR;
    float wreg[G::NW];
    /* The boundary between two chunks is instruction work that the matrix pipe waits for (f64 MFMA and VALU share the datapath:
     * tools/fir_timeline.py shows the SIMD's two waves taking turns, and every boundary instruction is a cycle the pipe idles),
     * so a window sample costs two VALU instructions to request (one masked byte offset, one load from a scalar base) and two to
     * deliver (v_cvt_f64_f32, v_cmp_class).  Round 2's first version spent ~25 on it and 2700 cycles per boundary. */
    const char *ringbytes = reinterpret_cast<const char *>(ringrow);
    const unsigned rmaskb = (unsigned)rmask << 2;
    auto win_fetch = [&](int s0, int ckc) {
        const int JT = (4 * (ckc - 1) + NR - 1) / NR;
        /* frame of (row, entry): F0 - 3 + NR (entry + 1 - 4 s0 / NR - JT) + row */
        const unsigned b0 = (unsigned)(a.ring.wpos + F0 - 3 + NR * (went + 1 - 4 * s0 / NR - JT) + wrow) << 2;
        /* ONE masked offset per lane and chunk; the lane's further samples lie 64 frames apart behind it -- immediates -- and run on
         * into the row's second copy where the ring wraps (Ring: every sample is stored twice, R floats apart) */
        if constexpr (LEAN) {
            const char *p0 = ringbytes + (b0 & rmaskb);
#pragma unroll
            for (int t = 0; t < G::NW; t++) wreg[t] = *reinterpret_cast<const float *>(p0 + 256 * t);
        } else {
#pragma unroll
            for (int t = 0; t < G::NW; t++) wreg[t] = *reinterpret_cast<const float *>(ringbytes + ((b0 + 256u * t) & rmaskb));
        }
    };
    auto win_store = [&](double *wdst) {
        double *wp = wdst + wrow * G::ROW + went;
        /* (double)sample IS the reference's operand (mulop) for every sample but Inf / NaN -- the wave runs with single-precision
         * subnormals flushed, so the conversion reads a subnormal as a signed zero, and the sign of a zero operand never reaches a sum
         * that starts at +0.  Inf and NaN are not looked for here (a v_cmp_class per sample was a sixth of a boundary's instructions):
         * they make the tile's sums non-finite, which is seen once, behind the last chunk, and such a tile is summed again the
         * reference's way (exact_tile). */
        bool odd = false;                                   /* (!LEAN) NaN, Inf or subnormal among the samples: the reference's bit-field operand */
#pragma unroll
        for (int t = 0; t < G::NW; t++)
            if (t * 64 + 63 < NR * G::ROW || lane + t * 64 < NR * G::ROW) {
                if constexpr (!LEAN) odd |= __builtin_amdgcn_classf(wreg[t], 0x297);
                wp[t * (64 / NR)] = (double)wreg[t];
            }
        if constexpr (!LEAN) {
            if (__builtin_expect(__ballot(odd) != 0, 0)) {      /* (no audio stream gets here) the same again with mulop() */
#pragma unroll
                for (int t = 0; t < G::NW; t++)
                    if (t * 64 + 63 < NR * G::ROW || lane + t * 64 < NR * G::ROW) wp[t * (64 / NR)] = mulop(wreg[t]);
            }
        }
    };
    /* A chunk's taps image u = 0 .. 4 ckc + 16 (R-1) + 27 <- Hbuf[48 - 16 (R-1) + 4 s0 + u] is a plain copy (the taps are doubles
     * already), so it goes by LDS-DMA: 1 KiB per instruction, lane l's 16 bytes to image + 16 l, no registers, nothing to wait for
     * until the image is read a chunk later.  (Through registers the copy stood in the way: requested at the boundary, its memory
     * latency was 3000 of a boundary's 4000 cycles, a fifth of a wave's life -- tools/fir_timeline.py; requested a chunk ahead, the
     * 24-36 registers it held slowed the k-step loop by more than that.) */
    /* The copy instruction is written by name: the lane's source = a SCALAR base + the lane's 32-bit offset, the destination's LDS
     * address in M0 from a scalar.  The compiler's own form (__builtin_amdgcn_global_load_lds) costs more than its address arithmetic:
     * it books the instruction as a FLAT access that may touch LDS, and while one is in flight -- here: all through the k-steps, by
     * design -- every wait for an LDS read becomes lgkmcnt(0), i.e. a k-step waits for the reads it has just issued instead of
     * those of two steps ago (six such waits per 16 k-steps; 66.2 cycles per MFMA at four row tiles where the pipe takes 64).
     * (M0 is the compiler's: saved and restored inside the statement.  These loads are not in the compiler's vmcnt arithmetic, which
     * only makes its own waits wait for more; the wait for a taps image is written out at the head of a chunk.) */
    auto dma16 = [&](const char *sbase, unsigned voff, unsigned lds_addr) __attribute__((always_inline)) {
        unsigned keep;
        asm volatile("s_mov_b32 %[k], m0\n\t"
                     "s_mov_b32 m0, %[d]\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %[v], %[s]\n\t"
                     "s_mov_b32 m0, %[k]"
                     : [k] "=&s"(keep) : [v] "v"(voff), [s] "s"(sbase), [d] "s"(lds_addr) : "memory");
    };
    const unsigned hs_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char *)hs);
    const unsigned lane16 = (unsigned)lane * 16u;
    auto taps_dma = [&](int which, int s0, int ckc) {
        const unsigned long long sv = reinterpret_cast<unsigned long long>(hbuf + (kTapsLead - 16) - 16 * (R - 1) + 4 * s0);
        const char *src = reinterpret_cast<const char *>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(sv >> 32)) << 32) |
                                                         (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)sv));    /* (the same in every lane: say so; the builtin returns int) */
        const int np = (4 * ckc + 16 * (R - 1) + 28) / 2;          /* 16-byte pieces the chunk needs; lanes beyond re-read the last one */
        const unsigned dst = hs_addr + (unsigned)which * (unsigned)(G::HLEN * 8);
        /* (LEAN, whole 1-KiB pieces: what the last one reads beyond the chunk's need lies inside the row's own zero tail -- kTapsTail
         * covers HLEN - HNEED < 128 doubles -- and lands behind the part of the image that is read) */
#pragma unroll
        for (int t = 0; t < G::HLEN / 128; t++)
            if (64 * t < np) {
                if constexpr (LEAN) dma16(src + 1024 * t, lane16, dst + 1024u * t);
                else {
                    /* the long boundary keeps the compiler's form ON PURPOSE: it is the one chosen where a cascade of the next block runs
                     * beside this launch (launch_fir), and that cascade lives on the bubbles the conservative waits leave (DESIGN.md 5b:
                     * with the waits exact a 512-chain shard's step goes 0.0870 -> 0.0983 ms, a 2048-chain one 0.2602 -> 0.2733) */
                    const int pi = lane + 64 * t;
                    __builtin_amdgcn_global_load_lds(src + 16 * (pi < np ? pi : np - 1), (__attribute__((address_space(3))) char *)(hs + which * G::HLEN) + 1024 * t, 16, 0, 0);
                }
            }
    };

    double q[16], bq[4];
    /* the queue as a chunk's first step finds it, and the operands of its first two steps
     * lane bases: taps u = (k + i) + 16 (R-1) + 4 sc; window row (3 - k) + ..., entry a + JT - ... (win_off) */
    auto chunk_begin = [&](const double *hp, const double *wp) {
#pragma unroll
        for (int j = 0; j < G::QD; j++) q[(16 - G::QD + j) & 15] = hp[4 * j];
        q[0] = hp[16 * (R - 1)]; q[1] = hp[16 * (R - 1) + 4];
        bq[0] = wp[win_off<R, BIG>(0)]; bq[1] = wp[win_off<R, BIG>(1)];
    };
    /* one k-step: the reads of step j + 2 (of the next group for j = 14, 15: the same code, offsets continue), R MFMAs */
    auto kstep = [&](const double *hg, const double *wg, auto jc) {
        constexpr int j = decltype(jc)::value;
        q[(j + 2) & 15] = hg[4 * (j + 2)];
        bq[(j + 2) & 3] = j + 2 < 16 ? wg[win_off<R, BIG>((j + 2) & 15)] : (wg - 64 / NR)[win_off<R, BIG>((j + 2) & 15)];
        __builtin_amdgcn_sched_barrier(0);          /* the reads stay two steps ahead of their MFMAs */
#pragma unroll
        for (int r = 0; r < R; r++)
            acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(q[(j - 4 * (R - 1 - r)) & 15], bq[j & 3], acc[r], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto group16 = [&](const double *hg, const double *wg) {
        kstep(hg, wg, std::integral_constant<int, 0>{});  kstep(hg, wg, std::integral_constant<int, 1>{});
        kstep(hg, wg, std::integral_constant<int, 2>{});  kstep(hg, wg, std::integral_constant<int, 3>{});
        kstep(hg, wg, std::integral_constant<int, 4>{});  kstep(hg, wg, std::integral_constant<int, 5>{});
        kstep(hg, wg, std::integral_constant<int, 6>{});  kstep(hg, wg, std::integral_constant<int, 7>{});
        kstep(hg, wg, std::integral_constant<int, 8>{});  kstep(hg, wg, std::integral_constant<int, 9>{});
        kstep(hg, wg, std::integral_constant<int, 10>{}); kstep(hg, wg, std::integral_constant<int, 11>{});
        kstep(hg, wg, std::integral_constant<int, 12>{}); kstep(hg, wg, std::integral_constant<int, 13>{});
        kstep(hg, wg, std::integral_constant<int, 14>{}); kstep(hg, wg, std::integral_constant<int, 15>{});
    };

    /* A chunk: write the window image (its samples were requested a chunk ago), request the next chunk's window samples and
     * taps image, run the k-steps.  While a wave is at a boundary the other wave of its SIMD has the matrix pipe to itself.
     * (Tried and dropped, measured slower: a second WINDOW image filled in slices between the MFMAs -- the slices' address
     * arithmetic and conversions cost the f64 matrix pipe more than the stop they replace, 83 -> 95 us on a 512-chain shard.) */
    win_fetch(Sb, min(ck, S - Sb));
    taps_dma(0, Sb, min(ck, S - Sb));
    [[maybe_unused]] int stamp_i = 1;
    int cur = 0;
    for (int s0 = Sb; s0 < S; s0 += ck, cur ^= 1) {
        const int ckc = min(ck, S - s0);
        const int JT = (4 * (ckc - 1) + NR - 1) / NR;
        __builtin_amdgcn_wave_barrier();
        FIR_STAMP_CHUNK();
        win_store(ws);                                      /* (waits for the window samples requested a chunk ago) */
        if (s0 == Sb + ck) FIR_STAMP(24);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    /* ... and for this chunk's taps image, in flight since then */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (s0 == Sb + ck) FIR_STAMP(25);
        if (s0 + ckc < S) {
            win_fetch(s0 + ckc, min(ck, S - s0 - ckc));
            if (s0 == Sb + ck) FIR_STAMP(26);
            taps_dma(cur ^ 1, s0 + ckc, min(ck, S - s0 - ckc));
        }
        if (s0 == Sb + ck) FIR_STAMP(27);
        const double *hp = hs + cur * G::HLEN + k + i16;                       /* oldest operand of step sc at hp[4 sc] */
        const double *wp = ws + (3 - k) * G::ROW + i16 + JT - (64 / NR - 1);   /* group g: wp - g * (64 / NR) + win_off(j) */
        chunk_begin(hp, wp);
        FIR_STAMP_CHUNK();
        for (int g = 0; g < ckc / 16; g++) group16(hp + 16 * (R - 1) + 64 * g, wp - g * (64 / NR));
        FIR_STAMP_CHUNK();
    }
    FIR_STAMP(23);
    /* C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg */
    if constexpr (LEAN) {
        /* Inf / NaN among the tile's samples (dspMulFloatDouble reads exponent 255 as 2^128 x 1.m, the matrix pipe as what it is): every
         * such sample reaches every output of the tile it is under, and Inf x 0 is a NaN, so the sums of a tile that met one are not
         * finite.  One look; such a tile -- no audio stream has one -- is summed again tap by tap with the reference's operands. */
        bool odd = false;
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int v = 0; v < 4; v++)
                odd |= F0 + NR * i16 + 16 * r + 4 * v + k < B && ((unsigned long long)__double_as_longlong(acc[r][v]) >> 52 & 0x7FF) == 0x7FF;
        if (__builtin_expect(__ballot(odd) != 0, 0)) {
            const float *ftaps = reinterpret_cast<const float *>(a.buf + c.fir_coef_word);
            const int Sfrom = SPLIT ? (half ? 4 * Sb - NR : 0) : 0, Sto = SPLIT && !half ? min(T, 4 * S - NR) : T;      /* (the tap split: this wave's taps) */
#pragma unroll
            for (int r = 0; r < R; r++)                     /* (unrolled: the sums stay in registers) */
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const int n = F0 + NR * i16 + 16 * r + 4 * v + k;
                    double sum = 0.0;
                    for (int tp = max(Sfrom, 0); tp < Sto; tp++)
                        sum = __builtin_fma(mulop(ringrow[(a.ring.wpos + n - tp) & rmask]), mulop(ftaps[tp]), sum);
                    acc[r][v] = sum;
                }
        }
    }
    if constexpr (R >= 2) {
        /* The tiles leave as 16-byte (R = 4) or 8-byte (R = 2) pieces.  A wave's results are one CHANNEL: stored by itself it
         * writes 4 bytes into each of 256 R different 128-byte lines, and with every wave of the chip doing that the L2's request
         * rate is what the epilogue waits for (~13 000 cycles per wave at R = 4, tools/fir_timeline.py).  The workgroup holds R
         * neighbouring channels over the same 1024 frames (4 / R tiles each): every wave parks its words in its own (now idle)
         * LDS region, [frame / NR][NR + 1] so that neither side has bank conflicts, and after one barrier every thread sends
         * four frames x R channels.  Only when the chains store once each to R consecutive, aligned output columns; anything else
         * goes the plain way. */
        constexpr int CH = R, WPC = G::WPC;                  /* channels per workgroup, waves (tiles) per channel -- of a 1024-frame block */
        const bool four = R == 2 && wsh == 0;                 /* (R = 2, one 512-frame tile per chain: the workgroup is FOUR channels) */
        unsigned *mine = reinterpret_cast<unsigned *>(hs);
        unsigned w16[4 * R];
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                unsigned word = store_stage<FMT>(acc[r][v], c.sat, a.io.store_mask);
                if constexpr (FMT == 6) word = ftz_bits(word);      /* default MODE here: flush the float by hand */
                w16[4 * r + v] = word;
                mine[(NR + 1) * i16 + 16 * r + 4 * v + k] = word;
            }
        if (lane == 0) xchg[wv] = c.n_out == 1 ? c.out_io[0] : -1;
        __syncthreads();
        const int o0 = xchg[0];
        const int chn = four ? 4 : CH;
        bool together = o0 >= 0 && ((o0 - a.io.out_base) & (chn - 1)) == 0 && (a.io.out_stride & (chn - 1)) == 0 &&
                        (reinterpret_cast<size_t>(a.io.out) & (4 * chn - 1)) == 0;
#pragma unroll
        for (int w = 1; w < 4; w++) together = together && xchg[w] == o0 + (w >> wsh);       /* wave w: channel w >> wsh, tile w & (tiles - 1) */
        if (together && four) {
            /* four channels, one tile of <= 512 frames each: 16-byte pieces like R = 4 */
            if constexpr (R == 2) {
                const unsigned *reg = reinterpret_cast<const unsigned *>(lds);
#pragma unroll
                for (int qf = 0; qf < 2; qf++) {
                    const int f = (int)threadIdx.x + 256 * qf;
                    if (f < B) {
                        const int at = (NR + 1) * (f / NR) + f % NR;
                        unsigned o[4];
#pragma unroll
                        for (int j = 0; j < 4; j++) o[j] = reg[(size_t)j * (2 * G::LDS_DOUBLES) + at];
                        unsigned *dst = a.io.out + (size_t)f * a.io.out_stride + (o0 - a.io.out_base);
                        *reinterpret_cast<uint4 *>(dst) = make_uint4(o[0], o[1], o[2], o[3]);
                    }
                }
            }
        } else if (together) {
            const unsigned *reg = reinterpret_cast<const unsigned *>(lds);
#pragma unroll
            for (int qf = 0; qf < 4; qf++) {
                const int f = (int)threadIdx.x + 256 * qf;
                if (f < B) {
                    const int ti = f / G::FW, fl = f - ti * G::FW;
                    const int at = (NR + 1) * (fl / NR) + fl % NR;
                    unsigned o[CH];
#pragma unroll
                    for (int j = 0; j < CH; j++) o[j] = reg[(size_t)(j * WPC + ti) * (2 * G::LDS_DOUBLES) + at];
                    unsigned *dst = a.io.out + (size_t)f * a.io.out_stride + (o0 - a.io.out_base);
                    if constexpr (CH == 4) *reinterpret_cast<uint4 *>(dst) = make_uint4(o[0], o[1], o[2], o[3]);
                    else *reinterpret_cast<uint2 *>(dst) = make_uint2(o[0], o[1]);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const int n = F0 + NR * i16 + 16 * r + 4 * v + k;
                    if (n < B) emit_out(a.io, c, n, w16[4 * r + v]);
                }
        }
    } else {
        if constexpr (SPLIT) {
            /* the second half's sums reach the first half's wave through the second's own (idle) LDS region; the first adds them to
             * its own -- (taps 0 .. S/2) + (taps S/2 ..), not the reference's order -- and stores */
            v4f64 *park = reinterpret_cast<v4f64 *>(lds + (size_t)(wv | 1) * G::LDS_DOUBLES);
            if (half) park[lane] = acc[0];
            __syncthreads();
            if (half) return;
            acc[0] += park[lane];
        }
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int n = F0 + NR * i16 + 16 * r + 4 * v + k;
                if (n < B) {
                    unsigned word = store_stage<FMT>(acc[r][v], c.sat, a.io.store_mask);
                    if constexpr (FMT == 6) word = ftz_bits(word);      /* default MODE here: flush the float by hand */
                    emit_out(a.io, c, n, word);
                }
            }
    }
    FIR_STAMP(30); FIR_REALTIME(28);
}


/* ------------------------------------------------------------------------------------------
 * fir_stream<FMT, R>: fir_tile's contraction (same tile, same operand order, same accumulators: see there) with NO
 * instruction work at the chunk boundaries.
 *
 * What tools/fir_timeline.py showed of fir_tile: a wave that streams independent f64 MFMAs starves the VALU instructions of
 * the other wave on its SIMD (v_mfma_f64 and VALU share the datapath and the MFMA stream wins the arbitration), so the ~100
 * VALU instructions a boundary spent converting and placing window samples took the whole of the partner's k-step phase,
 * the two waves ran their k-steps strictly in turns, and what was left of the boundary after the partner's last MFMA
 * (1100-1800 cycles per chunk) was idle matrix pipe: 4-6 % of the kernel at R = 4, plus the 70.8-cycle MFMA cadence of a
 * lone R = 1 wave.  Here
 *   - the window operand comes ready-made: every writer of the ring also writes mulop(sample) as a double (Ring::wide),
 *     so staging a chunk's window is a COPY, like its taps, and both go by LDS-DMA (global_load_lds, 1 KiB per
 *     instruction, no VGPR, no conversion, nothing to wait for until the chunk starts);
 *   - the copies of chunk c+1 are issued between the MFMAs of chunk c, three per group of 16 k-steps, into the other
 *     half of a double-buffered image; a boundary is s_waitcnt vmcnt(0) and the first four operand reads;
 *   - one wave per SIMD (a workgroup of four waves owns the CU's LDS): nothing shares the matrix pipe with it.
 * Window image: the B operand of lane (a, k) at k-step s is frame F0 + NR (a+1) - 4 s - k.  In g = frame + 3 the four k
 * of a step are an aligned quad, and the image is the ring's own order in blocks of NR doubles with 2 doubles of padding
 * behind each: position(g) = (g - glo) + 2 floor((g - glo) / NR).  Lane (a, k) reads base + (NR + 2) a + 3 - k, the 32
 * lanes of a ds_read_b64 pass hit 32 different bank pairs ((NR + 2) = 2 mod 32), and inside a group of 16 k-steps every
 * address is the group's pointer plus an immediate.  The DMA's lane l of piece p copies 16 bytes = image unit U = l + 64 p
 * = ring doubles glo + 2 (U - U / (NR/2 + 1)) and the next (the ring's g = frame + 3 numbering keeps pairs aligned and
 * inside the ring).
 * ---------------------------------------------------------------------------------------- */
template <int R> struct StreamGeom {
    static constexpr int NR = 16 * R;
    static constexpr int FW = 256 * R;                       /* frames per unit (one wave's output tile) */
    static constexpr int QD = 4 * (R - 1);
    static constexpr int PD = R == 4 ? 2 : 4;                /* k-steps the operand reads run ahead of their MFMAs (an LDS round trip) */
    static constexpr int CK = 160;                           /* k-steps per chunk, multiple of 16 */
    static constexpr int UB = NR / 2 + 1;                    /* 16-byte units per block of the window image */
    /* units a copy instruction advances the image by: whole blocks where several fit 64 lanes (the source of a lane is then
     * a per-lane constant plus a per-piece scalar; the lanes past the last whole block copy what the next piece's first
     * lanes copy, the same bytes to the same place), all 64 lanes at R = 4 (one division per lane and piece) */
    static constexpr int UP = R == 4 ? 64 : UB * (64 / UB);
    static constexpr int MAGIC = (65536 + UB - 1) / UB;      /* U / UB == (U * MAGIC) >> 16 for U < 1985 */
    static constexpr int GS = 64 + 128 / NR;                 /* doubles the window pointer moves per group of 16 k-steps */
    static constexpr int C60 = (60 + NR - 1) / NR;
    static constexpr int LOW = 60 + 2 * C60;
    __host__ __device__ static constexpr int wunits(int ckc) { return UB * (15 + 4 * ckc / NR) + 2; }
    __host__ __device__ static constexpr int wpieces(int ckc) { return (wunits(ckc) - 64 + UP - 1) / UP + 1; }    /* the last piece ends at or behind the last unit */
    __host__ __device__ static constexpr int hpieces(int ckc) { return (4 * ckc + 16 * (R - 1) + 28 + 4 * PD + 127) / 128; }
    static constexpr int WLEN = ((wpieces(CK) - 1) * UP + 64) * 2;   /* doubles of one window image */
    static constexpr int HLEN = hpieces(CK) * 128;                   /* doubles of one taps image */
    static constexpr int LDS_DOUBLES = 2 * HLEN + 2 * WLEN;          /* per wave */
    /* window offset (doubles, >= 0) of k-step j inside a group, from the group's pointer */
    __host__ __device__ static constexpr int woff(int j) { return 4 * (15 - j) + 2 * (C60 - (4 * j + NR - 1) / NR); }
};
static_assert(4 * StreamGeom<4>::LDS_DOUBLES * 8 + 64 <= 160 * 1024 && StreamGeom<4>::wpieces(StreamGeom<4>::CK) * 64 < 1985, "fir_stream: a CU's LDS holds four waves");
static_assert(4 * StreamGeom<2>::LDS_DOUBLES * 8 <= 160 * 1024 && 4 * StreamGeom<1>::LDS_DOUBLES * 8 <= 160 * 1024, "fir_stream: a CU's LDS holds four waves");

/* one unit = (chain, tile of FW frames); everything here is the same in all lanes of the wave */
struct StreamUnit {
    int cid, F0, S, ck;                 /* chain, first frame, k-steps (multiple of 16), k-steps per chunk */
    const double *hbuf;                 /* the chain's taps as doubles */
    const char *ring8;                  /* the chain's row of the operand ring */
};

constexpr int kStreamBlock = 512;       /* four consumer waves and four producer waves, one of each per SIMD */

template <int FMT, int R>
__global__ __launch_bounds__(kStreamBlock, 1) void fir_stream(const FirTileArgs a)
{
    using G = StreamGeom<R>;
    constexpr int NR = G::NR, PD = G::PD;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int B = a.io.nframes;
    const int tiles = (B + G::FW - 1) / G::FW;               /* units per chain */
    const int nunits = a.ngroup * tiles;
    const int npairs = gridDim.x * 4;

    /* Who is who.  The SIMD's two waves make a pair: the first to register consumes (MFMAs), the second produces (copies).
     * (Waves of a workgroup are dealt round the four SIMDs; should a workgroup ever come out uneven, wave w pairs with w + 4.) */
    int *ctl = reinterpret_cast<int *>(lds + (size_t)4 * G::LDS_DOUBLES);        /* [0..3] waves per SIMD, [4 + 2 pair] filled, [5 + 2 pair] consumed */
    if (threadIdx.x < 16) ctl[threadIdx.x] = 0;
    __syncthreads();
    const int simd = (__builtin_amdgcn_s_getreg((4 << 0) | (4 << 6) | (1 << 11))) & 3;      /* HW_ID.SIMD_ID */
    int myslot = 0;
    if (lane == 0) myslot = atomicAdd(&ctl[simd], 1);
    myslot = __builtin_amdgcn_readfirstlane(myslot);
    __syncthreads();
    const bool even = ctl[0] == 2 && ctl[1] == 2 && ctl[2] == 2 && ctl[3] == 2;
    const int pair = __builtin_amdgcn_readfirstlane(even ? simd : (wv & 3));
    const bool producer = __builtin_amdgcn_readfirstlane(even ? myslot : (wv >> 2)) != 0;
    int *filled = ctl + 4 + 2 * pair, *consumed = filled + 1;                    /* chunks copied / chunks multiplied, of the pair's sequence */

    int u = xcd_remap(blockIdx.x, a.per_xcd) * 4 + pair;     /* the pair's units: u, u + npairs, ... */
    if (u >= nunits) return;                                 /* (both waves of the pair; no barrier below) */
    double *hs = lds + (size_t)pair * G::LDS_DOUBLES, *ws = hs + 2 * G::HLEN;   /* taps images at hs, hs + HLEN; window images at ws, ws + WLEN */
    const unsigned rmask8 = (unsigned)(a.ring.R - 1) << 3;
    const int i16 = lane & 15, k = lane >> 4;

    auto unit_of = [&](int uu) {
        StreamUnit d;
        const int slot = uu / tiles;
        d.cid = __builtin_amdgcn_readfirstlane(a.group[slot]);
        d.F0 = (uu - slot * tiles) * G::FW;
        const int T = __builtin_amdgcn_readfirstlane(a.chains[d.cid].fir_taps);
        d.S = (((T + NR + 3) >> 2) + 15) & ~15;              /* k-steps: m_s = -NR + 4 s */
        const int nch = (d.S + G::CK - 1) / G::CK;
        d.ck = (((d.S + nch - 1) / nch) + 15) & ~15;
        d.hbuf = a.taps64 + (size_t)d.cid * a.pitch64;
        d.ring8 = reinterpret_cast<const char *>(wide_row(a.ring, d.cid));
        return d;
    };
    /* (bounded: the partner wave is resident in the same workgroup and always gets there; should that ever not hold, the grid
     * still drains after ~1 s of polling -- with wrong samples, which the parity tests would show -- instead of hanging the GPU) */
    auto flag_wait = [&](int *flag, int need) {
        for (int polls = 0; polls < (1 << 23); polls++) {
            if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) >= need) break;
            __builtin_amdgcn_s_sleep(4);
        }
    };
    auto flag_set = [&](int *flag, int v) {
        if (lane == 0) __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };

    if (producer) {
        /* ---- the copies: chunk n of the pair's sequence goes to image n & 1 once chunk n - 2 has been multiplied ---- */
        const unsigned lane16 = (unsigned)lane * 16u;
        constexpr int WPMAX = G::wpieces(G::CK);
        /* ring bytes of the lane's unit in piece p, from the image's first unit */
        unsigned rel8[R == 4 ? WPMAX : 1];
        if constexpr (R == 4) {
#pragma unroll
            for (int p = 0; p < WPMAX; p++) {
                const unsigned U = (unsigned)lane + 64u * (unsigned)p;
                rel8[p] = (2u * (U - ((U * (unsigned)G::MAGIC) >> 16))) << 3;
            }
        } else
            rel8[0] = (2u * ((unsigned)lane - (unsigned)lane / (unsigned)G::UB)) << 3;
        int n = 0;
        for (; u < nunits; u += npairs) {
            const StreamUnit d = unit_of(u);
            for (int s0 = 0; s0 < d.S; s0 += d.ck, n++) {
                const int ckc = min(d.ck, d.S - s0);
                const int npw = G::wpieces(ckc), nph = G::hpieces(ckc);
                if (n >= 2) flag_wait(consumed, n - 1);
                double *wdst = ws + (n & 1) * G::WLEN, *hdst = hs + (n & 1) * G::HLEN;
                const unsigned g08 = (unsigned)(a.ring.wpos + d.F0 + NR - 4 * (s0 + ckc)) << 3;      /* ring byte of image unit 0 */
#pragma unroll
                for (int p = 0; p < WPMAX; p++)
                    if (p < npw) {
                        unsigned off;
                        if constexpr (R == 4) off = (g08 + rel8[p]) & rmask8;
                        else off = (g08 + (unsigned)(8 * NR * (64 / G::UB) * p) + rel8[0]) & rmask8;    /* piece p starts at block p (64 / UB) */
                        __builtin_amdgcn_global_load_lds(d.ring8 + off, (__attribute__((address_space(3))) char *)(wdst + 2 * G::UP * p), 16, 0, 0);
                    }
                /* (the f64 taps carry kTapsTail zeros: a whole piece may be read) */
                const char *src = reinterpret_cast<const char *>(d.hbuf + (kTapsLead - 16) - 16 * (R - 1) + 4 * s0);
#pragma unroll
                for (int t = 0; t < G::hpieces(G::CK); t++)
                    if (t < nph)
                        __builtin_amdgcn_global_load_lds(src + 1024 * t + lane16, (__attribute__((address_space(3))) char *)(hdst + 128 * t), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                flag_set(filled, n + 1);
            }
        }
        return;
    }

    /* ---- the contraction ---- */
    [[maybe_unused]] const int stamp_row = pair;
    FIR_STAMP(0); FIR_REALTIME(29);
#ifdef AVDSP_FIR_STAMPS
    if (lane == 0) a.stamps[(size_t)(blockIdx.x * 4 + pair) * 32 + 31] = ((unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) << 32) |
                                                                         __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));    /* HW_ID | XCC_ID */
#endif
    v4f64 acc[R];
    double q[16], bq[8];
    auto chunk_begin = [&](const double *hp, const double *wg) {
#pragma unroll
        for (int j = 0; j < G::QD; j++) q[(16 - G::QD + j) & 15] = hp[4 * j];
#pragma unroll
        for (int j = 0; j < PD; j++) { q[j] = hp[16 * (R - 1) + 4 * j]; bq[j] = wg[G::woff(j)]; }
    };
    auto kstep = [&](const double *hg, const double *wg, auto jc) {
        constexpr int j = decltype(jc)::value;
        q[(j + PD) & 15] = hg[4 * (j + PD)];
        bq[(j + PD) & 7] = j + PD < 16 ? wg[G::woff((j + PD) & 15)] : (wg - G::GS)[G::woff((j + PD) & 15)];
        __builtin_amdgcn_sched_barrier(0);          /* the reads stay PD steps ahead of their MFMAs */
#pragma unroll
        for (int r = 0; r < R; r++)
            acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(q[(j - 4 * (R - 1 - r)) & 15], bq[j & 7], acc[r], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    [[maybe_unused]] int stamp_i = 1;
    int n = 0;
    for (; u < nunits; u += npairs) {
        const StreamUnit d = unit_of(u);
        const avdsp_chain c = a.chains[d.cid];
#pragma unroll
        for (int r = 0; r < R; r++) acc[r] = v4f64{0.0, 0.0, 0.0, 0.0};
        if (n == 0) FIR_STAMP(24);
        for (int s0 = 0; s0 < d.S; s0 += d.ck, n++) {
            const int ckc = min(d.ck, d.S - s0);
            FIR_STAMP_CHUNK();
            flag_wait(filled, n + 1);
            const double *hp = hs + (n & 1) * G::HLEN + k + i16;                                      /* oldest operand of step sc at hp[4 sc] */
            const double *wg = ws + (n & 1) * G::WLEN + (NR + 2) * i16 + 3 - k - G::LOW + (ckc / 16) * G::GS;    /* group 0; group g at wg - g GS */
            chunk_begin(hp, wg);
            FIR_STAMP_CHUNK();
            for (int g = 0; g < ckc / 16; g++) {
                const double *hg = hp + 16 * (R - 1) + 64 * g, *wgg = wg - g * G::GS;
                kstep(hg, wgg, std::integral_constant<int, 0>{});  kstep(hg, wgg, std::integral_constant<int, 1>{});
                kstep(hg, wgg, std::integral_constant<int, 2>{});  kstep(hg, wgg, std::integral_constant<int, 3>{});
                kstep(hg, wgg, std::integral_constant<int, 4>{});  kstep(hg, wgg, std::integral_constant<int, 5>{});
                kstep(hg, wgg, std::integral_constant<int, 6>{});  kstep(hg, wgg, std::integral_constant<int, 7>{});
                kstep(hg, wgg, std::integral_constant<int, 8>{});  kstep(hg, wgg, std::integral_constant<int, 9>{});
                kstep(hg, wgg, std::integral_constant<int, 10>{}); kstep(hg, wgg, std::integral_constant<int, 11>{});
                kstep(hg, wgg, std::integral_constant<int, 12>{}); kstep(hg, wgg, std::integral_constant<int, 13>{});
                kstep(hg, wgg, std::integral_constant<int, 14>{}); kstep(hg, wgg, std::integral_constant<int, 15>{});
            }
            FIR_STAMP_CHUNK();
            flag_set(consumed, n + 1);
        }
        /* C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg */
        const bool first_unit = n * 1 == (d.S + d.ck - 1) / d.ck;
        if (first_unit) FIR_STAMP(25);
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int nn = d.F0 + NR * i16 + 16 * r + 4 * v + k;
                if (nn < B) {
                    unsigned word = store_stage<FMT>(acc[r][v], c.sat, a.io.store_mask);
                    if constexpr (FMT == 6) word = ftz_bits(word);      /* default MODE here: flush the float by hand */
                    emit_out(a.io, c, nn, word);
                }
            }
        if (first_unit) FIR_STAMP(26);
    }
    FIR_STAMP(30); FIR_REALTIME(28);
}

/* ------------------------------------------------------------------------------------------
 * fir_flow<FMT, R> (round 4): fir_tile's launch -- one wave per (chain, tile), two waves per SIMD, nothing shared between waves, the
 * workgroup's tiles leaving together -- with fir_stream's operands: the window comes READY-MADE from the operand ring (every writer
 * of a ring leaves mulop(sample) there as a double, twice, R apart), so a chunk's window image is a copy like its taps image, both by
 * LDS-DMA, and a chunk boundary has NO vector instruction in it: the copy instructions take their lane's source from a register
 * computed once per wave and a scalar base, the rest is scalar arithmetic and a wait.
 *
 * Why that matters: while one wave of a SIMD streams MFMAs the other gets a VECTOR instruction in only every few hundred cycles (they
 * share the FP64 datapath), so fir_tile's boundary -- 140 vector instructions, 50 in its lean form -- lasts about as long as the
 * partner's chunk: the two waves take strict turns, the pipe runs one wave's dependent stream at a time (66 cycles per MFMA at four
 * row tiles, 72 at one) and idles for what is left of a boundary when the partner's chunk ends.  Scalar instructions, LDS-DMA
 * requests and waits are not held up by the partner's stream: this boundary is the copy's latency, under which the partner streams
 * alone, and the rest of the time BOTH waves stream (64.5 cycles per MFMA, r01's microbench).
 * The window image is single (the copy for chunk n + 1 is requested when chunk n's last k-step has read it: its latency is the
 * boundary), the taps images are two (requested a chunk ahead, as in fir_tile).  Same tile, same operand order, same accumulators as
 * fir_tile and fir_stream: bit-exact for the same reason, Inf / NaN included (the operand ring holds mulop(sample)).
 * ---------------------------------------------------------------------------------------- */
/* BIG (R = 1 only): launches that leave a SIMD one wave at most (fir_tile's BIG, same rule).  Nothing hides a boundary there, and the
 * CU's LDS is shared by four waves instead of eight: chunks of up to 224 k-steps and TWO window images, the copy of chunk n + 1's
 * window requested with its taps at the head of chunk n -- a boundary is then one s_waitcnt that has nothing left to wait for and
 * the first operand reads. */
template <int R, bool BIG = false> struct FlowGeom {
    static_assert(!BIG || R == 1, "long chunks and two window images: one row tile");
    static constexpr int NR = 16 * R;
    static constexpr int FW = 256 * R;
    static constexpr int WPC = 4 / R;
    static constexpr int QD = 4 * (R - 1);
    static constexpr int PD = R == 4 ? 2 : 4;
    static constexpr int CK = R == 4 ? 96 : R == 2 ? 128 : BIG ? 224 : 144;       /* k-steps per chunk: what leaves a CU's LDS to eight (BIG: four) waves */
    static constexpr int NWIN = BIG ? 2 : 1;
    static constexpr int UB = NR / 2 + 1;
    static constexpr int UP = R == 4 ? 64 : UB * (64 / UB);
    static constexpr int MAGIC = (65536 + UB - 1) / UB;
    static constexpr int GS = 64 + 128 / NR;
    static constexpr int C60 = (60 + NR - 1) / NR;
    static constexpr int LOW = 60 + 2 * C60;
    __host__ __device__ static constexpr int wunits(int ckc) { return UB * (15 + 4 * ckc / NR) + 2; }
    __host__ __device__ static constexpr int wpieces(int ckc) { return (wunits(ckc) - 64 + UP - 1) / UP + 1; }
    __host__ __device__ static constexpr int hpieces(int ckc) { return (4 * ckc + 16 * (R - 1) + 28 + 4 * PD + 127) / 128; }
    static constexpr int WLEN = ((wpieces(CK) - 1) * UP + 64) * 2;
    static constexpr int HLEN = hpieces(CK) * 128;
    static constexpr int LDS_DOUBLES = 2 * HLEN + NWIN * WLEN;       /* per wave: two taps images, one window image (BIG: two) */
    __host__ __device__ static constexpr int woff(int j) { return 4 * (15 - j) + 2 * (C60 - (4 * j + NR - 1) / NR); }
};
static_assert(8 * FlowGeom<4>::LDS_DOUBLES * 8 + 128 <= 160 * 1024 && 8 * FlowGeom<2>::LDS_DOUBLES * 8 + 128 <= 160 * 1024 &&
              8 * FlowGeom<1>::LDS_DOUBLES * 8 + 128 <= 160 * 1024, "fir_flow: a CU's LDS holds eight waves");
static_assert(4 * FlowGeom<1, true>::LDS_DOUBLES * 8 + 128 <= 160 * 1024, "fir_flow, long chunks: a CU's LDS holds four waves");
static_assert(FlowGeom<4>::wpieces(FlowGeom<4>::CK) * 64 < 1985, "fir_flow: the unit-to-block division by multiplication");

template <int FMT, int R, bool BIG = false>
__global__ __launch_bounds__(kBlock, BIG ? 1 : 2) void fir_flow(const FirTileArgs a)
{
    using G = FlowGeom<R, BIG>;
    constexpr int NR = G::NR, PD = G::PD;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      /* (scalar: the copies' bases and M0 come from it) */
    const int B = a.io.nframes;
    const int blk = xcd_remap(blockIdx.x, a.per_xcd);
    {   /* FIR-only chains: their input is appended first (fir_tile) */
        FirArgs fa{};
        fa.ring = a.ring; fa.io = a.io;
        for (int q = 0; q < 4 / G::WPC; q++) {
            const int sl = blk * (4 / G::WPC) + q;
            if (sl < a.ngroup) {
                const int ci = a.group[sl];
                const avdsp_chain cc = a.chains[ci];
                if (cc.nsec == 0) fir_append_input<FMT>(fa, cc, ci);
            }
        }
        __syncthreads();
    }
    const int unit = blk * 4 + wv;
    const int slot = unit / G::WPC, F0 = (unit % G::WPC) * G::FW;
    [[maybe_unused]] int *xchg = reinterpret_cast<int *>(lds + (size_t)4 * G::LDS_DOUBLES);
    [[maybe_unused]] const int stamp_row = wv;
    FIR_STAMP(0); FIR_REALTIME(29);
#ifdef AVDSP_FIR_STAMPS
    if (lane == 0) a.stamps[(size_t)(blockIdx.x * 4 + wv) * 32 + 31] = ((unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) << 32) |
                                                                        (unsigned long long)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 15);
#endif
    if (slot >= a.ngroup || F0 >= B) {
        if constexpr (R >= 2) {
            if (lane == 0) xchg[wv] = -1;
            __syncthreads();
        }
        return;
    }
    const int cid = __builtin_amdgcn_readfirstlane(a.group[slot]);
    const avdsp_chain c = a.chains[cid];
    const int T = __builtin_amdgcn_readfirstlane(c.fir_taps);
    if (a.ready && c.nsec) chain_ready_wait(a.ready, cid, a.seq, a.timeouts, a.ready_acquire != 0);
    double *hs = lds + (size_t)wv * G::LDS_DOUBLES, *ws = hs + 2 * G::HLEN;
    const double *hbuf = a.taps64 + (size_t)cid * a.pitch64;
    const char *ring8 = reinterpret_cast<const char *>(wide_row(a.ring, cid));
    const unsigned rmask8 = (unsigned)(a.ring.R - 1) << 3;
    const int i16 = lane & 15, k = lane >> 4;
    const int S = (((T + NR + 3) >> 2) + 15) & ~15;
    const int nch = (S + G::CK - 1) / G::CK;
    const int ck = (((S + nch - 1) / nch) + 15) & ~15;

    /* the copies: image unit U = lane + 64 p (16 bytes) <- ring doubles glo + 2 (U - U / UB) and the next; the lane's part of that is a
     * register made once (R = 4: one per piece), the chunk's part a scalar -- and since the ring holds every operand twice, R apart, the
     * masked start may simply be run on from */
    constexpr int WPMAX = G::wpieces(G::CK);
    unsigned rel8[R == 4 ? WPMAX : 1];
    if constexpr (R == 4) {
#pragma unroll
        for (int p = 0; p < WPMAX; p++) {
            const unsigned U = (unsigned)lane + 64u * (unsigned)p;
            rel8[p] = (2u * (U - ((U * (unsigned)G::MAGIC) >> 16))) << 3;
        }
    } else
        rel8[0] = (2u * ((unsigned)lane - (unsigned)lane / (unsigned)G::UB)) << 3;
    const unsigned lane16 = (unsigned)lane * 16u;
    /* One copy instruction, written by name: the lane's source = a SCALAR base + the lane's 32-bit offset (no vector add), the
     * destination's LDS address in M0 from a scalar (the compiler's own form reads it back out of a vector register for every piece).
     * (M0 is the compiler's: saved and restored inside the statement.  The loads are not in the compiler's count: the s_waitcnt at the
     * head of a chunk is written out.) */
    auto dma16 = [&](const char *sbase, unsigned voff, unsigned lds_addr) __attribute__((always_inline)) {
        unsigned keep;
        asm volatile("s_mov_b32 %[k], m0\n\t"
                     "s_mov_b32 m0, %[d]\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %[v], %[s]\n\t"
                     "s_mov_b32 m0, %[k]"
                     : [k] "=&s"(keep) : [v] "v"(voff), [s] "s"(sbase), [d] "s"(lds_addr) : "memory");
    };
    const unsigned ws_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char *)ws);
    const unsigned hs_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char *)hs);
    auto window_dma = [&](int which, int s0, int ckc) {
        const int npw = G::wpieces(ckc);
        const unsigned wdst = ws_addr + (unsigned)which * (G::WLEN * 8u);
        const unsigned g08 = ((unsigned)(a.ring.wpos + F0 + NR - 4 * (s0 + ckc)) << 3) & rmask8;       /* ring byte of image unit 0: scalar */
        const char *base = ring8 + g08;
#pragma unroll
        for (int p = 0; p < WPMAX; p++)
            if (p < npw) {
                if constexpr (R == 4) dma16(base, rel8[p], wdst + 16u * G::UP * p);
                else dma16(base + 8 * NR * (64 / G::UB) * p, rel8[0], wdst + 16u * G::UP * p);
            }
    };
    auto taps_dma = [&](int which, int s0, int ckc) {
        const int nph = G::hpieces(ckc);
        const char *src = reinterpret_cast<const char *>(hbuf + (kTapsLead - 16) - 16 * (R - 1) + 4 * s0);
#pragma unroll
        for (int t = 0; t < G::hpieces(G::CK); t++)
            if (t < nph) dma16(src + 1024 * t, lane16, hs_addr + (unsigned)which * (G::HLEN * 8u) + 1024u * t);
    };

    v4f64 acc[R];
#pragma unroll
    for (int r = 0; r < R; r++) acc[r] = v4f64{0.0, 0.0, 0.0, 0.0};
    double q[16], bq[8];
    auto chunk_begin = [&](const double *hp, const double *wg) {
#pragma unroll
        for (int j = 0; j < G::QD; j++) q[(16 - G::QD + j) & 15] = hp[4 * j];
#pragma unroll
        for (int j = 0; j < PD; j++) { q[j] = hp[16 * (R - 1) + 4 * j]; bq[j] = wg[G::woff(j)]; }
    };
    auto kstep = [&](const double *hg, const double *wg, auto jc) {
        constexpr int j = decltype(jc)::value;
        q[(j + PD) & 15] = hg[4 * (j + PD)];
        bq[(j + PD) & 7] = j + PD < 16 ? wg[G::woff((j + PD) & 15)] : (wg - G::GS)[G::woff((j + PD) & 15)];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < R; r++)
            acc[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(q[(j - 4 * (R - 1 - r)) & 15], bq[j & 7], acc[r], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };

    window_dma(0, 0, min(ck, S));
    taps_dma(0, 0, min(ck, S));
    int n = 0;
    [[maybe_unused]] int stamp_i = 1;
    for (int s0 = 0; s0 < S; s0 += ck, n++) {
        const int ckc = min(ck, S - s0);
        FIR_STAMP_CHUNK();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    /* this chunk's window (requested at the boundary; BIG: a chunk ago) and taps (a chunk ago) have landed */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (s0 + ckc < S) {                                 /* (land under this chunk's k-steps) */
            taps_dma((n + 1) & 1, s0 + ckc, min(ck, S - s0 - ckc));
            /* (Tried: these copies one piece per k-step instead of in a row here -- with one row tile every MFMA waits ~72 cycles for the
             * one before, a shadow a copy instruction might fit.  It does not: a piece costs 165 cycles there, its scalar address
             * arithmetic and branches included; k-steps 14.7 k -> 17.9 k cycles per chunk for 1.1 k saved at its head, cfg4 39.2 -> 43.4 us.) */
            if constexpr (BIG) window_dma((n + 1) & 1, s0 + ckc, min(ck, S - s0 - ckc));
        }
        const double *hp = hs + (n & 1) * G::HLEN + k + i16;
        const double *wg = ws + (BIG ? (n & 1) * G::WLEN : 0) + (NR + 2) * i16 + 3 - k - G::LOW + (ckc / 16) * G::GS;
        chunk_begin(hp, wg);
        FIR_STAMP_CHUNK();
        for (int g = 0; g < ckc / 16; g++) {
            const double *hg = hp + 16 * (R - 1) + 64 * g, *wgg = wg - g * G::GS;
            kstep(hg, wgg, std::integral_constant<int, 0>{});  kstep(hg, wgg, std::integral_constant<int, 1>{});
            kstep(hg, wgg, std::integral_constant<int, 2>{});  kstep(hg, wgg, std::integral_constant<int, 3>{});
            kstep(hg, wgg, std::integral_constant<int, 4>{});  kstep(hg, wgg, std::integral_constant<int, 5>{});
            kstep(hg, wgg, std::integral_constant<int, 6>{});  kstep(hg, wgg, std::integral_constant<int, 7>{});
            kstep(hg, wgg, std::integral_constant<int, 8>{});  kstep(hg, wgg, std::integral_constant<int, 9>{});
            kstep(hg, wgg, std::integral_constant<int, 10>{}); kstep(hg, wgg, std::integral_constant<int, 11>{});
            kstep(hg, wgg, std::integral_constant<int, 12>{}); kstep(hg, wgg, std::integral_constant<int, 13>{});
            kstep(hg, wgg, std::integral_constant<int, 14>{}); kstep(hg, wgg, std::integral_constant<int, 15>{});
        }
        /* the boundary: every read of the window image has been used (an MFMA waits for its operands); one image: the next window may land */
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        FIR_STAMP_CHUNK();
        if constexpr (!BIG) { if (s0 + ckc < S) window_dma(0, s0 + ckc, min(ck, S - s0 - ckc)); }
    }
    FIR_STAMP(23);
    /* the tile leaves as in fir_tile (C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg) */
    if constexpr (R >= 2) {
        constexpr int CH = R, WPC = G::WPC;
        unsigned *mine = reinterpret_cast<unsigned *>(hs);
        unsigned w16[4 * R];
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                unsigned word = store_stage<FMT>(acc[r][v], c.sat, a.io.store_mask);
                if constexpr (FMT == 6) word = ftz_bits(word);
                w16[4 * r + v] = word;
                mine[(NR + 1) * i16 + 16 * r + 4 * v + k] = word;
            }
        if (lane == 0) xchg[wv] = c.n_out == 1 ? c.out_io[0] : -1;
        __syncthreads();
        const int o0 = xchg[0];
        bool together = o0 >= 0 && ((o0 - a.io.out_base) & (CH - 1)) == 0 && (a.io.out_stride & (CH - 1)) == 0 &&
                        (reinterpret_cast<size_t>(a.io.out) & (4 * CH - 1)) == 0;
#pragma unroll
        for (int w = 1; w < 4; w++) together = together && xchg[w] == o0 + w / WPC;
        if (together) {
            const unsigned *reg = reinterpret_cast<const unsigned *>(lds);
#pragma unroll
            for (int qf = 0; qf < 4; qf++) {
                const int f = (int)threadIdx.x + 256 * qf;
                if (f < B) {
                    const int ti = f / G::FW, fl = f - ti * G::FW;
                    const int at = (NR + 1) * (fl / NR) + fl % NR;
                    unsigned o[CH];
#pragma unroll
                    for (int j = 0; j < CH; j++) o[j] = reg[(size_t)(j * WPC + ti) * (2 * G::LDS_DOUBLES) + at];
                    unsigned *dst = a.io.out + (size_t)f * a.io.out_stride + (o0 - a.io.out_base);
                    if constexpr (CH == 4) *reinterpret_cast<uint4 *>(dst) = make_uint4(o[0], o[1], o[2], o[3]);
                    else *reinterpret_cast<uint2 *>(dst) = make_uint2(o[0], o[1]);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int v = 0; v < 4; v++) {
                    const int nn = F0 + NR * i16 + 16 * r + 4 * v + k;
                    if (nn < B) emit_out(a.io, c, nn, w16[4 * r + v]);
                }
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int nn = F0 + NR * i16 + 16 * r + 4 * v + k;
                if (nn < B) {
                    unsigned word = store_stage<FMT>(acc[r][v], c.sat, a.io.store_mask);
                    if constexpr (FMT == 6) word = ftz_bits(word);
                    emit_out(a.io, c, nn, word);
                }
            }
    }
    FIR_STAMP(30); FIR_REALTIME(28);
}

/* FIR-only chains: the FIR's input is (float)X of the load stage; it is appended to the rings before fir_stream starts
 * (consecutive threads take consecutive chains: the reads of the interleaved block coalesce) */
template <int FMT>
__global__ __launch_bounds__(kBlock) void fir_feed(const FirTileArgs a)
{
    const long long total = (long long)a.ngroup * a.io.nframes;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int slot = (int)(g % a.ngroup), n = (int)(g / a.ngroup);
        const int cid = a.group[slot];
        const avdsp_chain c = a.chains[cid];
        if (c.nsec) continue;
        unsigned raw = a.io.in[(size_t)n * a.io.in_stride + (c.in_io - a.io.in_base)];
        if constexpr (FMT == 6) raw = ftz_bits(raw);
        ring_put(a.ring, cid, n, ftz_bits(narrow_stage<FMT>(load_stage<FMT>(raw, c.load_mode, c.gain_bits))));
    }
}

/* the f64 copy of a chain's taps, made once per plan: Hbuf[j] = mulop(h[j - kTapsLead]), zeros around */
struct Taps64Args { const int *buf; const avdsp_chain *chains; const int *group; double *taps64; int pitch64; };
__global__ __launch_bounds__(kBlock) void taps_to_f64(const Taps64Args a)
{
    const int cid = a.group[blockIdx.x];
    const avdsp_chain c = a.chains[cid];
    const float *taps = reinterpret_cast<const float *>(a.buf + c.fir_coef_word);
    double *dst = a.taps64 + (size_t)cid * a.pitch64;
    for (int j = threadIdx.x; j < a.pitch64; j += blockDim.x) {
        const int t = j - kTapsLead;
        dst[j] = (t >= 0 && t < c.fir_taps) ? mulop(taps[t]) : 0.0;
    }
}

/* cross-check path: one thread per output frame, the reference's loop verbatim (ascending taps),
 * operands straight from the ring and the program words (L1/L2)                                */
template <int FMT>
__global__ __launch_bounds__(kBlock) void fir_plain(const FirArgs a)
{
    if constexpr (FMT != 2) flush_f32_subnormals_like_the_reference();
    const int slot = xcd_remap(blockIdx.x, a.per_xcd);
    if (slot >= a.ngroup) return;
    const int cid = a.group[slot];
    const avdsp_chain c = a.chains[cid];
    const int T = c.fir_taps, B = a.io.nframes;
    if (c.nsec == 0) fir_append_input<FMT>(a, c, cid);
    const float *taps = reinterpret_cast<const float *>(a.buf + c.fir_coef_word);
    const float *ringrow = ring_row(a.ring, cid);
    const int rmask = a.ring.R - 1;
    for (int n = threadIdx.x; n < B; n += blockDim.x) {
        double acc = 0.0;
        for (int t = 0; t < T; t++)
            acc = __builtin_fma(mulop(ringrow[(a.ring.wpos + n - t) & rmask]), mulop(taps[t]), acc);
        emit_out(a.io, c, n, store_stage<FMT>(acc, c.sat, a.io.store_mask));
    }
}

/* ring <-> reference delay-line layout, one workgroup per FIR chain */
struct RingConvArgs {
    int *buf; const avdsp_chain *chains; const int *group; int ngroup; Ring ring;
};
__global__ __launch_bounds__(kBlock) void ring_to_state(const RingConvArgs a)
{
    const int cid = a.group[blockIdx.x];
    const avdsp_chain c = a.chains[cid];
    float *st = reinterpret_cast<float *>(a.buf + c.fir_state_word);
    for (int i = threadIdx.x; i < c.fir_taps; i += blockDim.x) st[i] = *ring_at(a.ring, cid, -1 - i);
}
__global__ __launch_bounds__(kBlock) void state_to_ring(const RingConvArgs a)
{
    const int cid = a.group[blockIdx.x];
    const avdsp_chain c = a.chains[cid];
    const float *st = reinterpret_cast<const float *>(a.buf + c.fir_state_word);
    for (int i = threadIdx.x; i < a.ring.R; i += blockDim.x) ring_put(a.ring, cid, -1 - i, i < c.fir_taps ? __float_as_uint(st[i]) : 0u);
}

/* the operand ring made from the float ring, when fir_stream is first asked for (from then on every ring writer keeps it up) */
__global__ __launch_bounds__(kBlock) void ring_widen(const RingConvArgs a)
{
    const int cid = a.group[blockIdx.x];
    double *row = wide_row(a.ring, cid);
    const float *src = ring_row(a.ring, cid);
    for (int i = threadIdx.x; i < a.ring.R; i += blockDim.x) { const double w = mulop(src[i]); row[(i + 3) & (a.ring.R - 1)] = w; row[((i + 3) & (a.ring.R - 1)) + a.ring.R] = w; }
}

/* chains with neither biquads nor FIR: LOAD -> [SAT0DB] -> STORE */
struct PassArgs {
    const avdsp_chain *chains;
    const int *group;
    int ngroup;
    BlockIO io;
};

template <int FMT>
__global__ __launch_bounds__(kBlock) void passthrough(const PassArgs a)
{
    if constexpr (FMT != 2) flush_f32_subnormals_like_the_reference();
    const long long total = (long long)a.ngroup * a.io.nframes;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
        const int slot = (int)(g % a.ngroup), n = (int)(g / a.ngroup);
        const avdsp_chain c = a.chains[a.group[slot]];
        auto X = load_stage<FMT>(a.io.in[(size_t)n * a.io.in_stride + (c.in_io - a.io.in_base)], c.load_mode, c.gain_bits);
        emit_out(a.io, c, n, store_stage<FMT>(X, c.sat, a.io.store_mask));
    }
}

/* ------------------------------------------------------------------------------------------
 * host sample formats -> s.31 words (linux/avdsp_plugin.c:103-121): pure byte traffic, HBM-bound.
 * Each thread converts four consecutive samples from whole 32-bit loads (S24_3LE: 3 words in, 4 out;
 * S16: 2 words in, 4 out) so that a wave reads and writes contiguous, dword-aligned spans; the ragged
 * end (and a misaligned source) is done sample by sample.
 * ---------------------------------------------------------------------------------------- */
struct UnpackArgs { const unsigned char *src; unsigned *dst; size_t n; };

__device__ __forceinline__ unsigned pcm24_at(const unsigned char *p) { return ((unsigned)p[0] << 8) | ((unsigned)p[1] << 16) | ((unsigned)p[2] << 24); }

template <int PCM>
__global__ __launch_bounds__(kBlock) void pcm_unpack(const UnpackArgs a)
{
    const size_t quads = ((reinterpret_cast<size_t>(a.src) & 3) == 0) ? a.n / 4 : 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned *w = reinterpret_cast<const unsigned *>(a.src);
    for (size_t q = t0; q < quads; q += stride) {
        uint4 o;
        if constexpr (PCM == AVDSP_PCM_S24_3LE) {
            const unsigned w0 = w[3 * q], w1 = w[3 * q + 1], w2 = w[3 * q + 2];
            o.x = w0 << 8;
            o.y = ((w0 >> 24) | (w1 << 8)) << 8;
            o.z = ((w1 >> 16) | (w2 << 16)) << 8;
            o.w = w2 & 0xFFFFFF00u;
        } else {
            const unsigned w0 = w[2 * q], w1 = w[2 * q + 1];
            o.x = w0 << 16; o.y = w0 & 0xFFFF0000u; o.z = w1 << 16; o.w = w1 & 0xFFFF0000u;
        }
        reinterpret_cast<uint4 *>(a.dst)[q] = o;
    }
    for (size_t i = quads * 4 + t0; i < a.n; i += stride) {
        if constexpr (PCM == AVDSP_PCM_S24_3LE) a.dst[i] = pcm24_at(a.src + 3 * i);
        else a.dst[i] = ((unsigned)a.src[2 * i] | ((unsigned)a.src[2 * i + 1] << 8)) << 16;
    }
}


/* ------------------------------------------------------------------------------------------
 * "tagoutput" of the ALSA plugin (linux/avdsp_plugin.c:133-137): the first output channel of a core carries, in bits
 * 8..15 of every sample, a count derived from the PREVIOUS sample's upper half -- a transport check for bit-perfect
 * playback.  sample' = (sample & 0xFFFF0000) | (prev & 0xFF00), prev = ((sample & 0xFFFF0000) >> 8) + 0x100: the carried
 * value depends on the previous frame's own sample only, so every frame of a block is independent given the block's
 * column; `prev` of the last frame is kept on the device for the next call (cores in program order, blocks in order).
 * ---------------------------------------------------------------------------------------- */
struct TagArgs { int *out; int stride, nframes; int *prev; };
__global__ __launch_bounds__(kBlock) void tag_column(const TagArgs a)
{
    __shared__ int first_prev;
    if (threadIdx.x == 0) first_prev = *a.prev;
    __syncthreads();
    int last = 0;
    bool have_last = false;
    /* one workgroup: frame n reads frame n-1's ORIGINAL sample, so originals are fetched before anything is stored */
    for (int n0 = 0; n0 < a.nframes; n0 += blockDim.x) {
        const int n = n0 + (int)threadIdx.x;
        int cur = 0, before = 0;
        if (n < a.nframes) {
            cur = a.out[(size_t)n * a.stride] & (int)0xFFFF0000;
            before = n ? ((a.out[(size_t)(n - 1) * a.stride] & (int)0xFFFF0000) >> 8) + 0x100 : first_prev;
        }
        __syncthreads();
        if (n < a.nframes) {
            a.out[(size_t)n * a.stride] = cur | (before & 0x0000FF00);
            if (n == a.nframes - 1) { last = (cur >> 8) + 0x100; have_last = true; }
        }
        __syncthreads();
    }
    if (have_last) *a.prev = last;
}

static constexpr int kStrandMaxOps = 512;         /* operations per strand (strand_lanes keeps the list in LDS) */
#include "avdsp_interp.inc"


/* ------------------------------------------------------------------------------------------
 * chain_lane<FMT>: the chain shape  LOAD | LOAD_GAIN -> BIQUADS* -> [FIR] -> [SAT0DB] -> STORE+  with ONE LANE PER CHAIN,
 * frames in order, all state in the mirror in the reference's own layout (the FIR's shifting delay line included).  It is
 * the parallel path of DSP_FORMAT 3 and 5 -- float accumulators and the truncating dspMulFloatFloat (dsp_ieee754.h:364-368),
 * arithmetic that MFMA and v_fma_f64 cannot reproduce -- built from the interpreter's own stage functions, so its results
 * are the interpreter's (goldens in all five models); a program of N channels runs N lanes wide instead of on one wave.
 * ---------------------------------------------------------------------------------------- */
struct LaneArgs {
    int *buf; const avdsp_chain *chains; const int *sec_coef, *sec_state; int nchains; BlockIO io;
    /* chains with a FIR: the cascade's outputs go to the chain's row of a sequence buffer, x[m] of the block at seq[row + hist + m]
     * with the delay line's content in front of it (m < 0), and fir_lane works from there; seq == nullptr: the FIR runs here, tap by
     * tap on the mirror's delay line (single frames) */
    unsigned *seq; int pitch, hist;
    int rows_take;                   /* chains of 1 .. 16 sections run on chain_rows */
    int lane_hw;                     /* chain_rows: the hardware's toward-zero products where the exponents allow (AVDSP_OPT_LANE_HW) */
};

template <int FMT>
__global__ __launch_bounds__(64) void chain_lane(const LaneArgs a)
{
    using namespace interp;
    using alu_t = typename M<FMT>::alu;
    flush_f32_subnormals_like_the_reference();
    const int cid = blockIdx.x * 64 + threadIdx.x;
    if (cid >= a.nchains) return;
    const avdsp_chain c = a.chains[cid];
    if (a.rows_take && c.nsec >= 1 && c.nsec <= 16) return;                    /* chain_rows' */
    if (a.seq && c.nsec == 0 && c.fir_taps) return;                            /* fir_lane_feed's: no cascade, every frame's input at once */
    const unsigned *inp = a.io.in + (c.in_io - a.io.in_base);
    for (int n = 0; n < a.io.nframes; n++) {
        const unsigned raw = inp[(size_t)n * a.io.in_stride];
        alu_t X;
        if constexpr (M<FMT>::smp_int) {                                        /* dsp_runtime.c:565-607 */
            if (c.load_mode == AVDSP_LOAD_GAIN) X = fmul<FMT>(int_to_float_scaled((int)raw, 31), __uint_as_float(c.gain_bits));
            else X = from_int_scaled<FMT>((int)raw, 31);
        } else {
            X = to_alu<FMT>(__uint_as_float(raw));
            if (c.load_mode == AVDSP_LOAD_GAIN) X *= to_alu<FMT>(__uint_as_float(c.gain_bits));
        }
        for (int s = 0; s < c.nsec; s++)                                        /* :827-849, section by section */
            X = biquads<FMT>(X, a.buf + a.sec_coef[c.sec_base + s], a.buf + a.sec_state[c.sec_base + s], 1, 0);
        if (c.fir_taps) {
            if (a.seq) { a.seq[(size_t)cid * a.pitch + a.hist + n] = __float_as_uint(to_sp<FMT>(X)); continue; }
            X = fir<FMT>(to_sp<FMT>(X), a.buf + c.fir_coef_word, a.buf + c.fir_state_word, c.fir_taps);   /* :928-969 */
        }
        if (c.sat) X = sat0db<FMT>(X);                                          /* :464-475 */
        unsigned word;                                                          /* :610-633 */
        if constexpr (M<FMT>::smp_int) word = (unsigned)(s31_from_float(X.v) & a.io.store_mask);
        else word = __float_as_uint(to_sp<FMT>(X));
        emit_out(a.io, c, n, word);
    }
}

/* chain_rows<FMT>: the cascades of the float-accumulator models with ONE LANE PER (CHAIN, SECTION) -- biquad_pipe's arrangement on
 * the interpreter's own section update (bq_step<FMT>: float sums of truncating products, ~150 instructions), four chains of up to 16
 * sections per wave.  Frame f reaches section s at step f + s; a section hands its output to the next lane with a DPP row shift;
 * the row's 16 lanes fetch and convert 16 frames at a time (one batch ahead, parked in LDS) and the section-0 lane takes one per
 * step; the last section's lane stores what chain_lane would (or feeds the sequence buffer of fir_lane).  ids: the chains with
 * 1 .. 16 sections; chain_lane keeps the others. */
template <int FMT>
__global__ __launch_bounds__(64) void chain_rows(const LaneArgs a, const int *ids, int nids)
{
    using namespace interp;
    using alu_t = typename M<FMT>::alu;
    flush_f32_subnormals_like_the_reference();
    __shared__ int xin[2][4][16];
    const int lane = threadIdx.x, row = lane >> 4, rp = lane & 15;
    const int gi = blockIdx.x * 4 + row;
    const bool valid = gi < nids;
    const int cid = ids[valid ? gi : nids - 1];
    const avdsp_chain c = a.chains[cid];
    const int S = c.nsec, B = a.io.nframes;
    /* rows are RIGHT-aligned (round 4): the chain's last section sits in lane 15 of its row whatever the section count, so that its
     * results reach the row's lanes with one dpp (row_newbcast:15) and leave sixteen frames at a time; rp is the lane's place in the
     * row -- the slot of the input batch it fetches -- and s its section (negative: none) */
    const int s = rp - (16 - S);
    const bool mine = valid && s >= 0;
    BqState<FMT> q; q.acc = 0; q.x1 = q.x2 = q.y1 = q.y2 = 0;
    int cw[5] = {0, 0, 0, 0, 0};
    int *stw = a.buf;
    if (mine) {
        stw = a.buf + a.sec_state[c.sec_base + s];
        bq_load<FMT>(q, stw);
        const int *cp = a.buf + a.sec_coef[c.sec_base + s];
        for (int k = 0; k < 5; k++) cw[k] = cp[k];
    }
    const unsigned *inp = a.io.in + (c.in_io - a.io.in_base);
    auto fetch = [&](int n) -> int {                     /* frame n as the first section's input word (dsp_runtime.c:565-607, :827-849) */
        if (!valid || n >= B) return 0;
        const unsigned raw = inp[(size_t)n * a.io.in_stride];
        alu_t X;
        if constexpr (M<FMT>::smp_int) {
            if (c.load_mode == AVDSP_LOAD_GAIN) X = fmul<FMT>(int_to_float_scaled((int)raw, 31), __uint_as_float(c.gain_bits));
            else X = from_int_scaled<FMT>((int)raw, 31);
        } else {
            X = to_alu<FMT>(__uint_as_float(raw));
            if (c.load_mode == AVDSP_LOAD_GAIN) X *= to_alu<FMT>(__uint_as_float(c.gain_bits));
        }
        return bq_input<FMT>(X);
    };
    /* The section update (dsp_biquadSTD.h:84-119 with a float accumulator: five dspMaccFloatFloat) with every operand of the software
     * product TAKEN APART ONCE -- the coefficients for the block, a sample when it enters the section (it is then multiplied three
     * times: as x, x1, x2) and an output when it is made (y1, y2): word m = mantissa with hidden bit and the sign in bit 31, word e =
     * biased exponent (a coefficient's less 127), a large negative number for exponent 0 (fir_lane).  prod() is dsp_ieee754.h:342-375's
     * integer arithmetic, literally.  Two NaNs add the SSE way in the reference (FF::operator+=): a product whose exponent field
     * filled up may read as one, so the largest exponent sum is tracked and a wave that met one runs its block AGAIN through the
     * interpreter's own bq_step (exact = true): stores and the sequence buffer are simply written twice, state leaves only at the end. */
    struct Apart { unsigned m; int e; };
    auto apart = [](unsigned w, int bias) -> Apart {
        const int e = (int)(w >> 23 & 255u);
        return Apart{(w & 0x807FFFFFu) | 0x800000u, e ? e - bias : -4096};
    };
    auto whole = [](Apart v) -> unsigned { return (v.m & 0x807FFFFFu) | (unsigned)(v.e == -4096 ? 0 : v.e) << 23; };
    Apart cf[5];
    for (int k = 0; k < 5; k++) cf[k] = apart((unsigned)cw[k], 127);
    const BqState<FMT> q0 = q;
    int emax = 0;
    auto macc = [&](float acc, Apart x, Apart cc) -> float {
        const int e = x.e + cc.e;
        const unsigned p = (unsigned)(((unsigned long long)(x.m & 0xFFFFFFu) * (cc.m & 0xFFFFFFu)) >> 22);
        const unsigned up = p >> 25 & 1u;
        const unsigned e9 = ((unsigned)e | ((x.m ^ cc.m) >> 23 & 0x100u)) + up;
        const unsigned word = ((p >> (1u + up)) & 0x7FFFFFu) | (e9 << 23);
        const float sum = acc + __uint_as_float(word);
        emax = max(emax, e);
        return e >= 1 ? sum : acc;
    };
    auto run = [&](auto exact_c) {
        constexpr bool exact = decltype(exact_c)::value;
        float acc = q.acc.v;
        Apart x1 = apart((unsigned)q.x1, 0), x2 = apart((unsigned)q.x2, 0), y1 = apart((unsigned)q.y1, 0), y2 = apart((unsigned)q.y2, 0);
        xin[0][row][rp] = fetch(rp);
        __syncthreads();
        int y = 0;
        const int steps = B + 15;                        /* the longest row's last frame leaves its 16th section at step B - 1 + 15 */
        for (int t0 = 0; t0 < steps; t0 += 16) {
            const int cur = (t0 >> 4) & 1;
            const int ahead = fetch(t0 + 16 + rp);       /* (in flight under the batch's steps) */
            for (int k = 0; k < 16; k++) {
                const int t = t0 + k;
                int x = __builtin_amdgcn_update_dpp(0, y, kRowShr1, 0xF, 0xF, false);      /* the section before, one step ago */
                if (s == 0) x = xin[cur][row][k];
                const int f = t - s;
                if (mine && f >= 0 && f < B) {
                    float outv;
                    if constexpr (exact) { y = bq_step<FMT>(q, x, cw); outv = q.acc.v; }
                    else {
                        const Apart xn = apart((unsigned)x, 0);
                        acc = macc(acc, xn, cf[0]); acc = macc(acc, x1, cf[1]); acc = macc(acc, x2, cf[2]);
                        acc = macc(acc, y1, cf[3]); acc = macc(acc, y2, cf[4]);
                        x2 = x1; x1 = xn; y2 = y1; y1 = apart(__float_as_uint(acc), 0);
                        y = __float_as_int(acc); outv = acc;
                    }
                    if (s == S - 1) {
                        alu_t X = FF(outv);
                        if (c.fir_taps && a.seq) a.seq[(size_t)cid * a.pitch + a.hist + f] = __float_as_uint(to_sp<FMT>(X));
                        else {
                            if (c.fir_taps) X = fir<FMT>(to_sp<FMT>(X), a.buf + c.fir_coef_word, a.buf + c.fir_state_word, c.fir_taps);
                            if (c.sat) X = sat0db<FMT>(X);
                            unsigned word;
                            if constexpr (M<FMT>::smp_int) word = (unsigned)(s31_from_float(X.v) & a.io.store_mask);
                            else word = __float_as_uint(to_sp<FMT>(X));
                            emit_out(a.io, c, f, word);
                        }
                    }
                }
            }
            xin[cur ^ 1][row][rp] = ahead;
            __syncthreads();                             /* (one wave: orders the row's writes against the section-0 lane's reads) */
        }
        if constexpr (!exact) { q.acc = FF(acc); q.x1 = (int)whole(x1); q.x2 = (int)whole(x2); q.y1 = (int)whole(y1); q.y2 = (int)whole(y2); }
    };
    /* Round 4: the same steps with the HARDWARE's products (fir_lane_hw has the why and the probe): v_mul_f32 under round-toward-zero
     * is dspMulFloatFloat bit for bit while both exponent fields are 1 .. 254 and sum to 128 .. 380, and the five products of a step
     * do not wait for its sums, so a step is one switch to toward-zero, five products, one switch back, five adds -- 12 instructions
     * where macc() above takes ~75.  What may go that way is decided by exponents again, with the wave's coefficients' range
     * (cmin .. cmax over the non-zero exponent fields) taken once and every VALUE that is ever multiplied -- the first section's
     * inputs as they are staged, every section's outputs as they are made, the state the block starts from -- held against
     *     0 or an exponent field >= 153 - cmin   (products >= 2^-101: IEEE's, and every partial sum a multiple of 2^-124, so no sum is
     *                                            ever flushed to -0.0 -- x + (+-0) is x for every x but -0.0, and a product with a zero
     *                                            operand is +-0 here where the reference leaves the accumulator alone)
     *     and an exponent field <= min(254, 380 - cmax);
     * the accumulator a lane starts from must itself be +0 or >= 2^-101.  A wave that sees anything else -- before the block or
     * in it: the look is four instructions a step -- runs the block (again) the integer way from the untouched state. */
    auto run_hw = [&](unsigned lo_bits, unsigned hi_bits) -> bool {
        float acc = q.acc.v;
        float x1 = __int_as_float(q.x1), x2 = __int_as_float(q.x2), y1 = __int_as_float(q.y1), y2 = __int_as_float(q.y2);
        const float c0 = __int_as_float(cw[0]), c1 = __int_as_float(cw[1]), c2 = __int_as_float(cw[2]), c3 = __int_as_float(cw[3]), c4 = __int_as_float(cw[4]);
        auto out_of_band = [&](unsigned w) { const unsigned mag = w & 0x7FFFFFFFu; return (mag - 1u < lo_bits - 1u) | (mag >= hi_bits); };
        bool bad = false;
        const int first = fetch(rp);
        bad |= out_of_band((unsigned)first);
        xin[0][row][rp] = first;
        __syncthreads();
        int y = 0;
        const int steps = B + 15;
        /* where the last section's lane leaves frame f = t - s: the FIR's sequence buffer, or (one STORE, no FIR on the delay line) the
         * output column -- both as a pointer that every lane advances by a frame per step; anything else the general way */
        const bool is_last = mine && s == S - 1;
        const bool to_seq = c.fir_taps && a.seq;
        const bool plain_out = !c.fir_taps && c.n_out == 1;
        unsigned *op = to_seq ? a.seq + (size_t)cid * a.pitch + a.hist - s
                              : a.io.out + (c.out_io[0] - a.io.out_base) - (ptrdiff_t)s * a.io.out_stride;
        const size_t ostep = to_seq ? 1 : (size_t)a.io.out_stride;
        /* ... and, in the batches where every lane is busy, sixteen frames at a time: lane rp of the row takes the last section's result
         * of the batch's step rp (row_newbcast:15) and converts and stores frame t0 + rp - (S - 1) when the batch is through */
        unsigned *bp = to_seq ? a.seq + (size_t)cid * a.pitch + a.hist + (rp - (S - 1))
                              : a.io.out + (c.out_io[0] - a.io.out_base) + (ptrdiff_t)(rp - (S - 1)) * a.io.out_stride;
        auto leave_word = [&](float v, int f, unsigned *at) __attribute__((always_inline)) {
            alu_t X = FF(v);
            if (to_seq) *at = __float_as_uint(to_sp<FMT>(X));
            else {
                if (c.sat) X = sat0db<FMT>(X);
                unsigned word;
                if constexpr (M<FMT>::smp_int) word = (unsigned)(s31_from_float(X.v) & a.io.store_mask);
                else word = __float_as_uint(to_sp<FMT>(X));
                if (plain_out) *at = word; else emit_out(a.io, c, f, word);
            }
        };
        const bool batch_out = to_seq || !c.fir_taps;    /* (a FIR on the delay line -- single frames only -- keeps the frame-by-frame way) */
        auto leave = [&](int f) __attribute__((always_inline)) {
            alu_t X = FF(acc);
            if (to_seq) *op = __float_as_uint(to_sp<FMT>(X));
            else {
                if (c.fir_taps) X = fir<FMT>(to_sp<FMT>(X), a.buf + c.fir_coef_word, a.buf + c.fir_state_word, c.fir_taps);
                if (c.sat) X = sat0db<FMT>(X);
                unsigned word;
                if constexpr (M<FMT>::smp_int) word = (unsigned)(s31_from_float(X.v) & a.io.store_mask);
                else word = __float_as_uint(to_sp<FMT>(X));
                if (plain_out) *op = word; else emit_out(a.io, c, f, word);
            }
        };
        /* one section update: five products toward zero, five sums to nearest (dsp_biquadSTD.h:84-119 with a float accumulator) */
        auto update = [&](int x) __attribute__((always_inline)) {
            const float xn = __int_as_float(x);
            float p0, p1, p2, p3, p4;
            asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t"
                         "v_mul_f32 %[p0], %[x], %[c0]\n\t" "v_mul_f32 %[p1], %[x1], %[c1]\n\t" "v_mul_f32 %[p2], %[x2], %[c2]\n\t"
                         "v_mul_f32 %[p3], %[y1], %[c3]\n\t" "v_mul_f32 %[p4], %[y2], %[c4]\n\t"
                         "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\t"
                         "v_add_f32 %[acc], %[acc], %[p0]\n\t" "v_add_f32 %[acc], %[acc], %[p1]\n\t" "v_add_f32 %[acc], %[acc], %[p2]\n\t"
                         "v_add_f32 %[acc], %[acc], %[p3]\n\t" "v_add_f32 %[acc], %[acc], %[p4]\n\t"
                         : [acc] "+v"(acc), [p0] "=&v"(p0), [p1] "=&v"(p1), [p2] "=&v"(p2), [p3] "=&v"(p3), [p4] "=&v"(p4)
                         : [x] "v"(xn), [x1] "v"(x1), [x2] "v"(x2), [y1] "v"(y1), [y2] "v"(y2),
                           [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2), [c3] "v"(c3), [c4] "v"(c4));
            x2 = x1; x1 = xn; y2 = y1; y1 = acc;
            y = __float_as_int(acc);
            bad |= out_of_band((unsigned)y);
        };
        for (int t0 = 0; t0 < steps; t0 += 16) {
            const int cur = (t0 >> 4) & 1;
            const int ahead = fetch(t0 + 16 + rp);
            bad |= out_of_band((unsigned)ahead);
            /* the row's sixteen inputs of this batch, in registers (no LDS round trip in front of a step's products) */
            int xa[16];
            {
                typedef int i4 __attribute__((ext_vector_type(4)));
                const i4 *xp = reinterpret_cast<const i4 *>(&xin[cur][row][0]);
#pragma unroll
                for (int j = 0; j < 4; j++) { const i4 v = xp[j]; xa[4 * j] = v[0]; xa[4 * j + 1] = v[1]; xa[4 * j + 2] = v[2]; xa[4 * j + 3] = v[3]; }
            }
            if (t0 >= 16 && t0 + 15 < B) {
                /* every section of every row is busy in all sixteen steps: nothing to ask.  (Lanes without a section run along on zeros.) */
                int mine_out = 0;
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    int x = __builtin_amdgcn_update_dpp(0, y, kRowShr1, 0xF, 0xF, false);
                    if (s == 0) x = xa[k];
                    update(x);
                    if (batch_out) {
                        int yb;
                        asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_newbcast:15 row_mask:0xf bank_mask:0xf" : "=v"(yb) : "v"(y));
                        mine_out = rp == k ? yb : mine_out;
                    } else if (is_last) leave(t0 + k - s);
                    op += ostep;
                }
                if (batch_out && valid) leave_word(__int_as_float(mine_out), t0 + rp - (S - 1), bp + (size_t)t0 * ostep);
            } else {
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    int x = __builtin_amdgcn_update_dpp(0, y, kRowShr1, 0xF, 0xF, false);
                    if (s == 0) x = xa[k];
                    const int f = t0 + k - s;
                    if (mine && f >= 0 && f < B) {
                        update(x);
                        if (is_last) leave(f);
                    }
                    op += ostep;
                }
            }
            xin[cur ^ 1][row][rp] = ahead;
            __syncthreads();
        }
        if (__ballot(bad) != 0) return false;
        q.acc = FF(acc); q.x1 = __float_as_int(x1); q.x2 = __float_as_int(x2); q.y1 = __float_as_int(y1); q.y2 = __float_as_int(y2);
        return true;
    };
    bool done = false;
    if (a.lane_hw) {
        /* the wave's coefficient range (non-zero exponent fields), and the state the block starts from against the band it gives */
        unsigned cmin1 = 0xFFFFFFFFu, cmax = 0;
        if (mine)
            for (int k = 0; k < 5; k++) { const unsigned e = (unsigned)cw[k] >> 23 & 255u; cmin1 = min(cmin1, e - 1u); cmax = max(cmax, e); }
        for (int d = 1; d < 64; d <<= 1) { cmin1 = min(cmin1, (unsigned)__shfl_xor((int)cmin1, d, 64)); cmax = max(cmax, (unsigned)__shfl_xor((int)cmax, d, 64)); }
        if (cmax <= 254u) {
            const unsigned cmin = cmin1 == 0xFFFFFFFFu ? 255u : cmin1 + 1u;       /* no non-zero coefficient: every product is a zero */
            const unsigned lo_e = cmin >= 152u ? 1u : 153u - cmin, hi_e = min(254u, 380u - cmax);
            const unsigned lo_bits = lo_e << 23, hi_bits = (hi_e + 1u) << 23;
            bool bad0 = false;
            if (mine) {
                auto oob = [&](unsigned w) { const unsigned mag = w & 0x7FFFFFFFu; return (mag - 1u < lo_bits - 1u) | (mag >= hi_bits); };
                const unsigned ab = __float_as_uint(q.acc.v);
                bad0 = oob((unsigned)q.x1) | oob((unsigned)q.x2) | oob((unsigned)q.y1) | oob((unsigned)q.y2) |
                       (ab != 0u && ((ab & 0x7FFFFFFFu) < (26u << 23) || (ab & 0x7F800000u) == 0x7F800000u));
            }
            if (lo_e <= hi_e && __ballot(bad0) == 0) done = run_hw(lo_bits, hi_bits);
            if (!done) q = q0;
        }
    }
    if (!done) {
        run(std::false_type{});
        if (__builtin_expect(__ballot(emax >= 254) != 0, 0)) { q = q0; run(std::true_type{}); }
    }
    if (mine) bq_store<FMT>(q, stw);
}

/* fir_lane<FMT>: the DSP_FIR of the float-accumulator models, ONE LANE PER (CHAIN, FRAME).  An output is a sum over the chain's
 * inputs only -- no feedback -- so the frames of a block do not wait for each other; what must stay is the order inside one output:
 * acc = acc + dspMulFloatFloat(x[n-i], c[i]) for i = 0, 1, ... (dsp_firSTD.h:38-52 with the truncating product of
 * dsp_ieee754.h:342-375 and a float sum), which is what every lane does.  A workgroup takes 256 consecutive frames of one chain and
 * walks the taps in chunks of kFirLaneChunk: the chunk's coefficients and the 256 + chunk - 1 inputs under them are staged in
 * LDS (consecutive lanes read consecutive words, the coefficient is a broadcast).  SAT0DB and the stores follow as in chain_lane.
 * fir_lane_history lays the delay line (st[i] = x[-1-i]) in front of the block's inputs, fir_lane_state leaves it as the
 * reference's tap loop would (st[i] = x[B-1-i]). */
constexpr int kFirLaneChunk = 2048, kFirLaneFrames = 256;

__global__ __launch_bounds__(256) void fir_lane_history(const LaneArgs a)
{
    const int cid = blockIdx.x;
    const avdsp_chain c = a.chains[cid];
    const int i = blockIdx.y * 256 + threadIdx.x;
    if (i < c.fir_taps - 1) a.seq[(size_t)cid * a.pitch + a.hist - 1 - i] = (unsigned)a.buf[c.fir_state_word + i];
}

/* chains that are a FIR and nothing in front of it: the block's inputs, converted (dsp_runtime.c:565-607), straight into the sequence
 * buffer, a lane per (chain, frame) */
template <int FMT>
__global__ __launch_bounds__(256) void fir_lane_feed(const LaneArgs a)
{
    using namespace interp;
    using alu_t = typename M<FMT>::alu;
    flush_f32_subnormals_like_the_reference();
    const int cid = blockIdx.x, n = blockIdx.y * 256 + threadIdx.x;
    const avdsp_chain c = a.chains[cid];
    if (c.nsec != 0 || c.fir_taps == 0 || n >= a.io.nframes) return;
    const unsigned raw = a.io.in[(size_t)n * a.io.in_stride + (c.in_io - a.io.in_base)];
    alu_t X;
    if constexpr (M<FMT>::smp_int) {
        if (c.load_mode == AVDSP_LOAD_GAIN) X = fmul<FMT>(int_to_float_scaled((int)raw, 31), __uint_as_float(c.gain_bits));
        else X = from_int_scaled<FMT>((int)raw, 31);
    } else {
        X = to_alu<FMT>(__uint_as_float(raw));
        if (c.load_mode == AVDSP_LOAD_GAIN) X *= to_alu<FMT>(__uint_as_float(c.gain_bits));
    }
    a.seq[(size_t)cid * a.pitch + a.hist + n] = __float_as_uint(to_sp<FMT>(X));
}

__global__ __launch_bounds__(256) void fir_lane_state(const LaneArgs a)
{
    const int cid = blockIdx.x;
    const avdsp_chain c = a.chains[cid];
    const int i = blockIdx.y * 256 + threadIdx.x;
    if (i < c.fir_taps) a.buf[c.fir_state_word + i] = (int)a.seq[(size_t)cid * a.pitch + a.hist + a.io.nframes - 1 - i];
}

template <int FMT>
__global__ __launch_bounds__(kFirLaneFrames) void fir_lane(const LaneArgs a)
{
    using namespace interp;
    using alu_t = typename M<FMT>::alu;
    flush_f32_subnormals_like_the_reference();
    /* Both operands of dspMulFloatFloat are taken apart ONCE, while they are staged (an input is multiplied by 256 x 2048 taps
     * out of one workgroup's image, a tap by 256 frames): word 0 = the mantissa with its hidden bit in bits 0..23 and the sign in
     * bit 31 (the 24-bit multiplies do not look above bit 23), word 1 = the biased exponent -- the tap's already less 127 -- or a
     * large negative number for exponent 0, which sends the sum of the two below 1 like the reference's "operand or product too
     * small: the accumulator stays" (interp::fmacc).  What is left per tap is dsp_ieee754.h:342-375's integer arithmetic, literally:
     * exponent sum, sign into bit 8, 24 x 24 mantissa product >> 22, one normalising shift with its exponent step, pack, float add. */
    __shared__ uint2 ws[kFirLaneChunk + kFirLaneFrames], cs[kFirLaneChunk];
    const int cid = blockIdx.x, t = threadIdx.x, n0 = blockIdx.y * kFirLaneFrames;
    const avdsp_chain c = a.chains[cid];
    const int T = c.fir_taps;
    if (T == 0) return;                                  /* (uniform: the whole workgroup is one chain's) */
    const unsigned *x = a.seq + (size_t)cid * a.pitch + a.hist;          /* x[m], m = -(T-1) .. nframes-1 */
    auto apart = [](unsigned w, int bias) -> uint2 {
        const int e = (int)(w >> 23 & 255u);
        return make_uint2((w & 0x807FFFFFu) | 0x800000u, (unsigned)(e ? e - bias : -4096));
    };
    float acc = 0.0f;
    for (int i0 = 0; i0 < T; i0 += kFirLaneChunk) {
        const int tc = min(kFirLaneChunk, T - i0);
        /* ws[k] = x[n0 - i0 - (tc - 1) + k], k < tc + 255: lane t's tap i0 + ii reads ws[t + tc - 1 - ii] */
        const int base = n0 - i0 - (tc - 1);
        __syncthreads();
        for (int k = t; k < tc + kFirLaneFrames - 1; k += kFirLaneFrames) {
            const int m = base + k;
            ws[k] = apart(m < a.io.nframes ? x[m] : 0u, 0);
        }
        for (int k = t; k < tc; k += kFirLaneFrames) cs[k] = apart((unsigned)a.buf[c.fir_coef_word + i0 + k], 127);
        __syncthreads();
        const uint2 *w = ws + t + tc - 1;
        const float acc0 = acc;
        int emax = 0;
#pragma unroll 4
        for (int ii = 0; ii < tc; ii++) {
            const uint2 xv = w[-ii], cv = cs[ii];
            const int e = (int)xv.y + (int)cv.y;
            const unsigned p = (unsigned)(((unsigned long long)(xv.x & 0xFFFFFFu) * (cv.x & 0xFFFFFFu)) >> 22);
            const unsigned up = p >> 25 & 1u;                       /* the product reached 2: one more shift, one more in the exponent */
            const unsigned e9 = ((unsigned)e | ((xv.x ^ cv.x) >> 23 & 0x100u)) + up;
            const unsigned word = ((p >> (1u + up)) & 0x7FFFFFu) | (e9 << 23);
            const float sum = acc + __uint_as_float(word);
            acc = e >= 1 ? sum : acc;
            emax = max(emax, e);
        }
        /* a product whose exponent field filled up can read as a NaN, and two NaNs add the SSE way in the reference (FF::operator+=):
         * such a chunk -- no audio gets there -- is summed again through the interpreter's own fmacc on the words put back together */
        if (__builtin_expect(__ballot(emax >= 254) != 0, 0)) {
            alu_t A = FF(acc0);
            auto whole = [](uint2 v, int bias) { const int e = (int)v.y; return (v.x & 0x807FFFFFu) | (unsigned)(e == -4096 ? 0 : e + bias) << 23; };
            for (int ii = 0; ii < tc; ii++) A = fmacc<FMT>(A, __uint_as_float(whole(w[-ii], 0)), __uint_as_float(whole(cs[ii], 127)));
            acc = A.v;
        }
    }
    const int n = n0 + t;
    if (n >= a.io.nframes) return;
    alu_t X = FF(acc);
    if (c.sat) X = sat0db<FMT>(X);                                              /* :464-475 */
    unsigned word;                                                              /* :610-633 */
    if constexpr (M<FMT>::smp_int) word = (unsigned)(s31_from_float(X.v) & a.io.store_mask);
    else word = __float_as_uint(to_sp<FMT>(X));
    emit_out(a.io, c, n, word);
}

/* fir_lane_hw<FMT> (round 4): fir_lane with the HARDWARE's product wherever the hardware gives the reference's bits.
 *
 * dspMulFloatFloat (dsp_ieee754.h:335-375) is the exact 24 x 24-bit mantissa product cut to 24 bits: for operands with biased
 * exponents 1 .. 254 whose exponents sum to 128 .. 380 that IS v_mul_f32 under MODE.FP_ROUND = toward zero, bit for bit
 * (tools/rtz_mul_probe.hip: 0 of 535 747 pairs on every boundary of mantissa and carry differ; what differs is everything the
 * reference does not treat the IEEE way -- exponent sums of 127 and less (the reference looks before the carry and returns +0, the
 * hardware returns a signed zero or, on a carry at 127, the smallest normal number), exponent fields that overflow, an exponent of
 * 255 read as 2^128, +0 for a zero operand whatever the other is).  The adds stay at round-to-nearest: the mode is switched by
 * s_setreg_imm32_b32 around groups of products (the switch is taken at once and costs 2 % at two per 32 instructions, same probe;
 * a v_mul_f64 + v_cvt_f32_f64 pair under the DOUBLE round field does not work: the conversion obeys the single field).  Every
 * floating-point instruction of the tap loop sits inside asm statements between its two switches -- the compiler knows nothing of
 * the mode and must not find an add to move across one.
 *
 * Which chunks may take that path is decided from the OPERANDS' exponents, once per chunk and workgroup, while they are staged:
 * with xmin / xmax the smallest non-zero and the largest exponent field among the chunk's inputs and hmin / hmax the taps',
 *     xmin + hmin >= 153   every product of two non-zero-exponent operands has an exponent field >= 26: it is IEEE's (>= 128 is what
 *                          that takes), and every partial sum is a multiple of 2^-124 -- never subnormal, so never flushed to -0.0,
 *                          which matters because a product with a zero-exponent operand is +-0 here and "the accumulator stays" in
 *                          the reference (interp::fmacc): x + (+-0) is x for every x but -0.0;
 *     xmax + hmax <= 380, xmax, hmax <= 254   no exponent field fills up, no Inf / NaN operand.
 * Audio passes (a 24-bit LSB is exponent 103, taps down to 1e-15 are 77); a chunk that does not -- decays into the last few
 * hundred dB, Inf / NaN, huge values -- is summed by the interpreter's own fmacc, tap by tap, as fir_lane's replay does.
 *
 * A lane sums TWO adjacent frames (2 t, 2 t + 1 of the workgroup's 512): their inputs at taps (e, e + 1) are one aligned pair of
 * words P(e) = (x[f - e], x[f + 1 - e]) and the neighbours' halves -- x_A(e + 1) = P(e + 2).hi, x_B(e + 1) = P(e).lo -- so one
 * ds_read_b64 serves four products; the taps are wave-uniform and come by scalar loads.  Per tap and frame: one v_mul_f32 and one
 * v_add_f32 (fir_lane: 17 instructions). */
constexpr int kFirHwFrames = 512;
template <int FMT>
__global__ __launch_bounds__(256) void fir_lane_hw(const LaneArgs a)
{
    using namespace interp;
    using alu_t = typename M<FMT>::alu;
    flush_f32_subnormals_like_the_reference();
    constexpr int kFront = 56;                           /* zeros below ws[0]: the chunk's taps are padded to a multiple of 32 with zeros, whose inputs are read from there */
    __shared__ __attribute__((aligned(16))) unsigned wsbuf[kFront + kFirLaneChunk + kFirHwFrames + 8];
    __shared__ __attribute__((aligned(16))) unsigned hs[kFirLaneChunk + 32];
    unsigned *ws = wsbuf + kFront;
    __shared__ int rng[5];                               /* xmin - 1 (unsigned: a zero exponent counts as 2^32 - 1), xmax, hmin - 1, hmax; [4]: a lane's sum so far is -0.0 or under 2^-101 */
    const int cid = blockIdx.x, t = threadIdx.x, n0 = blockIdx.y * kFirHwFrames;
    const avdsp_chain c = a.chains[cid];
    const int T = c.fir_taps;
    if (T == 0) return;                                  /* (uniform: the whole workgroup is one chain's) */
    const unsigned *x = a.seq + (size_t)cid * a.pitch + a.hist;          /* x[m], m = -(T-1) .. nframes-1 */
    const unsigned *taps = reinterpret_cast<const unsigned *>(a.buf + c.fir_coef_word);
    float accA = 0.0f, accB = 0.0f;                      /* frames n0 + 2 t and n0 + 2 t + 1 */
    for (int i0 = 0; i0 < T; i0 += kFirLaneChunk) {
        const int tc = min(kFirLaneChunk, T - i0);
        /* ws[k] = x[base + k]; E = tc - 1 + pad is even, so that lane t's pair at an even tap e, (ws[2 t + E - e], ws[2 t + E - e + 1]),
         * is 8-byte aligned: base = n0 - i0 - E */
        const int E = (tc - 1 + 1) & ~1;
        const int base = n0 - i0 - E, nws = E + kFirHwFrames + 2;
        __syncthreads();
        if (t < 5) rng[t] = (t & 1) || t == 4 ? 0 : -1;  /* (min slots start at the unsigned maximum) */
        if (t < kFront) wsbuf[t] = 0u;
        __syncthreads();
        unsigned lo = 0xFFFFFFFFu, hi = 0, hlo = 0xFFFFFFFFu, hhi = 0;
        for (int k = t; k < nws; k += 256) {
            const int m = base + k;
            const unsigned w = (m < a.io.nframes && m >= n0 - i0 - (tc - 1)) ? x[m] : 0u;      /* (below: the padding word no tap reads) */
            ws[k] = w;
            const unsigned e = w >> 23 & 255u;
            lo = min(lo, e - 1u); hi = max(hi, e);
        }
        for (int k = t; k < ((tc + 31) & ~31); k += 256) {
            const unsigned w = k < tc ? taps[i0 + k] : 0u;
            hs[k] = w;
            const unsigned e = w >> 23 & 255u;
            hlo = min(hlo, e - 1u); hhi = max(hhi, e);
        }
        atomicMin(reinterpret_cast<unsigned *>(&rng[0]), lo); atomicMax(reinterpret_cast<unsigned *>(&rng[1]), hi);
        atomicMin(reinterpret_cast<unsigned *>(&rng[2]), hlo); atomicMax(reinterpret_cast<unsigned *>(&rng[3]), hhi);
        {   /* what an earlier chunk left (a chunk summed the integer way may leave anything): the multiple-of-2^-124 argument needs
             * sums that are +0 or at least 2^-101 to start from */
            const unsigned ba = __float_as_uint(accA), bb = __float_as_uint(accB);
            if ((ba != 0u && (ba & 0x7FFFFFFFu) < (26u << 23)) || (bb != 0u && (bb & 0x7FFFFFFFu) < (26u << 23))) rng[4] = 1;
        }
        __syncthreads();
        const unsigned xmin1 = (unsigned)rng[0], xmax = (unsigned)rng[1], hmin1 = (unsigned)rng[2], hmax = (unsigned)rng[3];
        /* (no non-zero exponent on one side: every product is a zero) */
        const bool hw_ok = rng[4] == 0 && xmax <= 254u && hmax <= 254u && xmax + hmax <= 380u &&
                           (xmin1 == 0xFFFFFFFFu || hmin1 == 0xFFFFFFFFu || xmin1 + hmin1 + 2u >= 153u);
        const unsigned *wl = ws + 2 * t + E;             /* lane t: x_A(e) = wl[-e], x_B(e) = wl[1 - e] */
        if (__builtin_expect(!hw_ok, 0)) {
            alu_t A = FF(accA), Bv = FF(accB);
            for (int e = 0; e < tc; e++) {
                const float h = __uint_as_float(taps[i0 + e]);
                A = fmacc<FMT>(A, __uint_as_float(wl[-e]), h);
                Bv = fmacc<FMT>(Bv, __uint_as_float(wl[1 - e]), h);
            }
            accA = A.v; accB = Bv.v;
            continue;
        }
        /* four taps per statement: pairs P(e), P(e + 2), P(e + 4); products toward zero, sums to nearest, tap order per frame.  The
         * taps are wave-uniform: four of them per broadcast read of the LDS image. */
#define AVDSP_FIRHW_4(P0L, P0H, P1L, P1H, P2H, H) \
        asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\t" \
                     "v_mul_f32 %[a0], %[h0], %[p0l]\n\t" "v_mul_f32 %[b0], %[h0], %[p0h]\n\t" \
                     "v_mul_f32 %[a1], %[h1], %[p1h]\n\t" "v_mul_f32 %[b1], %[h1], %[p0l]\n\t" \
                     "v_mul_f32 %[a2], %[h2], %[p1l]\n\t" "v_mul_f32 %[b2], %[h2], %[p1h]\n\t" \
                     "v_mul_f32 %[a3], %[h3], %[p2h]\n\t" "v_mul_f32 %[b3], %[h3], %[p1l]\n\t" \
                     "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\t" \
                     "v_add_f32 %[A], %[A], %[a0]\n\t" "v_add_f32 %[B], %[B], %[b0]\n\t" \
                     "v_add_f32 %[A], %[A], %[a1]\n\t" "v_add_f32 %[B], %[B], %[b1]\n\t" \
                     "v_add_f32 %[A], %[A], %[a2]\n\t" "v_add_f32 %[B], %[B], %[b2]\n\t" \
                     "v_add_f32 %[A], %[A], %[a3]\n\t" "v_add_f32 %[B], %[B], %[b3]\n\t" \
                     : [A] "+v"(accA), [B] "+v"(accB), [a0] "=&v"(q0), [b0] "=&v"(q1), [a1] "=&v"(q2), [b1] "=&v"(q3), \
                       [a2] "=&v"(q4), [b2] "=&v"(q5), [a3] "=&v"(q6), [b3] "=&v"(q7) \
                     : [p0l] "v"(P0L), [p0h] "v"(P0H), [p1l] "v"(P1L), [p1h] "v"(P1H), [p2h] "v"(P2H), \
                       [h0] "v"(H.x), [h1] "v"(H.y), [h2] "v"(H.z), [h3] "v"(H.w))
        /* (the image's taps beyond tc are zeros -- up to a multiple of 32 -- and their inputs come from the zeros below ws[0].)  The next
         * group's operands are read while this group's products and sums run (with two waves per SIMD nothing else hides an LDS round
         * trip); two groups per turn, so that the two register sets swap roles instead of being copied. */
        uint2 P[9], Pn[9];
        uint4 H[4], Hn[4];
        auto read16 = [&](uint2 (&p)[9], uint4 (&h)[4], int e) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < 9; j++) p[j] = *reinterpret_cast<const uint2 *>(wl - e - 2 * j);
#pragma unroll
            for (int j = 0; j < 4; j++) h[j] = *reinterpret_cast<const uint4 *>(hs + e + 4 * j);
        };
        auto sum16 = [&](const uint2 (&p)[9], const uint4 (&h)[4]) __attribute__((always_inline)) {
            float q0, q1, q2, q3, q4, q5, q6, q7;
            AVDSP_FIRHW_4(p[0].x, p[0].y, p[1].x, p[1].y, p[2].y, h[0]);
            AVDSP_FIRHW_4(p[2].x, p[2].y, p[3].x, p[3].y, p[4].y, h[1]);
            AVDSP_FIRHW_4(p[4].x, p[4].y, p[5].x, p[5].y, p[6].y, h[2]);
            AVDSP_FIRHW_4(p[6].x, p[6].y, p[7].x, p[7].y, p[8].y, h[3]);
        };
        read16(P, H, 0);
        for (int e = 0; e < tc; e += 32) {
            read16(Pn, Hn, e + 16);
            sum16(P, H);
            read16(P, H, e + 32 < tc ? e + 32 : e);           /* (behind the last pair: anything inside the images) */
            sum16(Pn, Hn);
        }
#undef AVDSP_FIRHW_4
    }
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const int n = n0 + 2 * t + half;
        if (n >= a.io.nframes) break;
        alu_t X = FF(half ? accB : accA);
        if (c.sat) X = sat0db<FMT>(X);                                          /* :464-475 */
        unsigned word;                                                          /* :610-633 */
        if constexpr (M<FMT>::smp_int) word = (unsigned)(s31_from_float(X.v) & a.io.store_mask);
        else word = __float_as_uint(to_sp<FMT>(X));
        emit_out(a.io, c, n, word);
    }
}

/* dspRuntime_N, one frame per call: the host's wait for the frame.  hipDeviceSynchronize() costs a trip through the runtime's completion signal
 * (an interrupt and a wake-up) per call; this one-thread kernel behind the frame's kernels writes the call's number into pinned host memory
 * instead -- the kernels in front of it have completed, their stores into the same pinned area are visible -- and the host looks at that
 * word in its own memory (avdsp_hip_run_block_host). */
__global__ void frame_done(unsigned *flag, unsigned seq)
{
    __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

/* ------------------------------------------------------------------------------------------
 * host side of the thin ABI
 * ---------------------------------------------------------------------------------------- */
struct Plan {
    bool generic = false;             /* general interpreter instead of chain kernels */
    GenericArgs ga{};                 /* launch template of the generic path (io filled per block) */
    int io_span = 0;                  /* highest IO number the core touches + 1 */
    bool dither_only = false;         /* a DSP_TPDF_CALC and nothing else: tpdf_walk */
    int dither_arg = 0, dither_word = 0;
    size_t ga_lds = 0; bool ga_staged = false;
    int format = 0, nchains = 0, store_mask = -1;
    int instances = 1;                /* > 1: the chains are that many copies of a core's chains (avdsp_plan_desc::instances) */
    avdsp_chain *d_chains = nullptr;
    int *d_sec_coef = nullptr, *d_sec_state = nullptr;
    /* launch groups (device arrays of chain ids) */
    struct Group { int P; int nsec; int n; int *d_ids; bool all_fir;        /* all_fir: every chain of the group feeds a FIR (its cascade writes the ring) */
                   RowRec *d_rows; LaneRec *d_lanes;                        /* biquad_row's records (P == 16) */
                   /* more than 64 sections (round 5): the group as pieces of up to 64, launched one after the other; piece k hands the
                    * word between its last section and piece k + 1's first through column j (the chain's place in the group) of
                    * d_scratch[k & 1], [1024 frames][n] words (kLoadRaw / kStoreRaw) */
                   std::vector<Group> pieces = {}; unsigned *d_scratch[2] = {nullptr, nullptr};
                   bool raw_out = false; };      /* a piece but the last: it stores its last section's result word as it is (biquad_row<4> can) */
    /* round 5: the rows of ALL the plan's 16-lane groups in one table -- runs of one section count, each filled up to whole waves (four
     * rows) with empty rows, the table to whole workgroups -- for ONE biquad_row launch instead of one per section count (nsec 0 in its
     * arguments: every wave takes its rows' count from their records).  Only made when there are two such groups or more. */
    RowRec *d_rows_all = nullptr; LaneRec *d_lanes_all = nullptr; int n_rows_all = 0; bool rows_all_fir = true; int n_row_groups = 0;
    std::vector<Group> bq;            /* biquad chains grouped by section count (P = lanes per chain) */
    int *d_fir_ids = nullptr;  int n_fir = 0, max_taps = 0;
    int *d_pass_ids = nullptr; int n_pass = 0;
    /* FIR history rings: [nchains][ring_R] floats, frame 0 of the next block goes to index wpos */
    float *d_ring = nullptr; int ring_R = 0, wpos = 0;
    double *d_ring64 = nullptr;                          /* fir_stream: the ring as window operands (Ring::wide) */
    int n_fir_only = 0;                                  /* FIR chains without a cascade in front */
    int fir_gpc = 0;                  /* groups of 16 tap positions per LDS chunk */
    int io_in_min = 0, io_in_max = -1, io_out_min = 0, io_out_max = -1;
    bool wave_ok = false; unsigned carried_io[8] = {0, 0, 0, 0, 0, 0, 0, 0};      /* frame-parallel interpreter */
    int *d_own = nullptr;                                /* owned mirror ranges (pairs), generic plans */
    double *d_taps64 = nullptr; int pitch64 = 0;         /* fir_tile: the taps as doubles, [chain][pitch64] */
    bool lane_mode = false;                              /* formats 3 and 5: chain_lane, one lane per chain, state in the mirror */
    int *d_lane_rows = nullptr; int n_lane_rows = 0;     /* ... chain_rows' chains (1 .. 16 sections) */
    int n_lane_feed = 0;                                 /* ... chains that are a FIR alone (fir_lane_feed) */
    int n_lane_fir = 0; unsigned *d_lseq = nullptr; int lseq_pitch = 0;      /* ... and fir_lane's sequence buffer: [nchains][max_taps - 1 + frames of the largest block so far] */
    /* strand plan attached to a generic plan (include/avdsp_hip.h): the stretch as micro-operations, one argument row per strand */
    avdsp_strand_op *d_sops = nullptr; int *d_sargs = nullptr; int s_nops = 0, s_nargs = 0, s_nstrands = 0, s_nres = 0; bool s_usey = false;
    std::vector<int> s_loaded, s_stored;                 /* the IOs the strands load / store: per call against the windows */
    bool stores_whole_window = false;                    /* every IO of [io_out_min, io_out_max] is stored by some chain */
    bool overlap_ok = false;                             /* every cascade of the plan feeds a FIR: its launches may run under the previous block's FIR */
    unsigned *d_ready = nullptr; unsigned seq = 0;       /* [nchains] ready words (chain_ready_*): the number of the latest launch whose cascade is through, and the launch counter */
};

}  // namespace

struct avdsp_hip_prog {
    int total_words = 0;
    int chain_inst = 0;                 /* > 1: d_buf holds that many copies of the mirror side by side, [instance][total_words] -- the instances of a program whose
                                           cores are chain cores (avdsp_hip_chain_instances): an instance is a further block of chains with its own state words */
    int *d_buf = nullptr;
    TpdfGlobals *d_tpdf = nullptr;
    int *d_tpdf_seq = nullptr; int tpdf_seq_frames = 0;   /* per-frame dither values inside a core cut into pieces */
    unsigned *d_frame = nullptr; int frame_words = 0;     /* samples[] frame of the general interpreter */
    std::vector<Plan> plans;
    unsigned *d_in = nullptr, *d_out = nullptr; size_t in_cap = 0, out_cap = 0;   /* host-call staging */
    struct Alias { hipStream_t stream; unsigned *buf; size_t cap; };
    std::vector<Alias> alias;           /* copy of the input block of an in-place device call (avdsp_hip_run_block), one per caller's stream: stream order is what keeps a copy until its block's kernels have read it */
    /* N instances of the program (avdsp_hip_run_levels_instances): copies 1 .. N-1 of the device state; instance 0 is the program's own */
    int inst_n = 1; bool inst_valid = false;
    bool ring_wait_host = true;              /* "ring_wait": the host (1), not the cascades' stream (0), waits for the FIR three blocks back (launch_all) */
    bool call_shown = false;                 /* inside a dspRuntimeBlockAll call whose shared window columns have been copied (show_through) */
    int *d_inst_buf = nullptr; TpdfGlobals *d_inst_tpdf = nullptr; unsigned *d_inst_frame = nullptr; int *d_inst_seq = nullptr;
    int inst_frame_words = 0, inst_seq_frames = 0;
    static constexpr int kSmallWords = 4096;             /* host calls of up to that many sample words (dspRuntime_N) ... */
    unsigned *h_small = nullptr, *d_small = nullptr;     /* ... go through a pinned area the kernels access in place */
    unsigned *done_offer = nullptr; bool done_taken = false;      /* ... offered to the frame's own kernel where that is ONE wave (interp_core) */
    unsigned small_seq = 0;                               /* ... whose last 16 words hold the "frame done" word a kernel behind the frame's kernels sets (frame_done) */
    /* optional per-kernel timing with HIP events on the launch stream (avdsp_hip_profile_*) */
    int num_cus = 0;                    /* compute units of the device (fir_stream's grid) */
    unsigned profile = 0;               /* bit k: time the launches of kind k (AVDSP_KERNEL_*) */
    int profile_stride = 1;             /* ... every profile_stride-th of them (an event pair costs the stream ~5 us) */
    unsigned profile_seen[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    struct Span { int kind; hipEvent_t a, b; bool pair; };   /* pair: two events recorded around the launch (reads ~5 us more than the kernel took), else the dispatch's own stamps */
    int last_read_pairs[8] = {0, 0, 0, 0, 0, 0, 0, 0};       /* avdsp_hip_profile_read: how many of the spans it summed were event pairs */
    std::vector<Span> spans;            /* recorded, not yet read */
    std::vector<hipEvent_t> free_events;
    /* cores of one level side by side (avdsp_hip_run_levels): side streams and their fork / join events */
    /* argument tables of grid launches (avdsp_hip_run_levels): a small ring of pinned host / device slot pairs, a slot
     * is reused once the launch that read it has finished */
    static constexpr int kTableSlots = 4;
    GenericArgs *h_table = nullptr, *d_table = nullptr; int table_cap = 0, table_next = 0;
    hipEvent_t table_done[kTableSlots] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<hipStream_t> side;
    std::vector<hipEvent_t> join;
    hipEvent_t fork = nullptr;
    /* cascade of block k+1 under the FIR of block k ("overlap"): the cascades run on a stream of their own */
    int overlap = 0;
    int fir_rows = 0;                    /* fir_tile: row tiles per wave (1, 2, 4), 0 = by the number of chains */
    int fir_lean = -1;                   /* fir_tile's lean chunk boundary: -1 by the plan (launch_fir), 0 never, 1 always */
    int fir_split = 0;                   /* fir_tile: launches of at most a tile per SIMD cut every tile's taps over two waves (sums within 1e-6, not the reference's bits) */
    hipStream_t s_bq = nullptr;
    hipStream_t s_fir[2] = {nullptr, nullptr};           /* "overlap" 2: the FIRs of consecutive blocks in turn */
    static constexpr int kAhead = 3;     /* cascade k waits for FIR k - kAhead: it may run under FIR k - 2 and be done before FIR k - 1 ends */
    hipEvent_t ev_bq[kAhead] = {nullptr, nullptr, nullptr}, ev_fir[kAhead] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fir_now[kAhead] = {nullptr, nullptr, nullptr};   /* what stands for "FIR of this slot has ended": ev_fir[slot], or the stop event of the kernel timer that rode on that launch (no second event on the stream) */
    hipEvent_t last_ride_stop = nullptr; /* ProfileScope::ride: the stop event the latest timed launch carries */
    /* How an un-timed FIR launch of the overlap mode is enqueued (launch_all), by what was measured:
     *   1  the event the next cascades wait for rides on the dispatch's own completion signal (hipExtLaunchKernel's stop slot) instead
     *      of a marker packet behind it: 4096 chains 0.5110 -> 0.5078 ms per step, 2048 chains 0.2592 -> 0.2572;
     *   2  the dispatch carries a start and a stop event (a marker in front of it, a completion signal of its own): where the
     *      cascade of the next blocks is what the step waits for -- FIR launches under ~0.2 ms: 512 chains 0.0971 -> 0.0868 ms,
     *      1024 chains 0.1559 -> 0.1464, cfg5's 2048-chain shard 0.1575 -> 0.1477 -- the few microseconds the queue spends on the
     *      marker are microseconds the starved cascade has the chip to itself; on the long launches it only costs (0.5078 -> 0.5202);
     *   0  a plain launch and a recorded event (round 3).
     * -1 (default): 2 for plans of at most 6 M taps in all (chains x taps), else 1.  AVDSP_OPT_FIR_LAUNCH / AVDSP_FIR_LAUNCH_MODE. */
    int fir_launch_mode = -1, fir_mode_now = 0;
    hipEvent_t launch_ev[16] = {};       /* ... mode 2: a small ring of start / stop event pairs */
    unsigned launch_ev_next = 0;
    int lane_hw = 1;                     /* formats 3 / 5: the hardware's toward-zero product where it is the reference's (fir_lane_hw, chain_rows' fast steps); 0: the integer products throughout */
    int ready_words = -1;                /* "overlap": how the FIR of a block finds its cascades' block in the rings: 0 an event between the two queues, 1 ready words published by
                                            the cascade's waves (slower everywhere), 2 ready words set by a kernel behind the cascade; -1 (default): 2 where the FIR is the
                                            bound, else 0 (launch_all) */
    int ready_mode_now = 0;              /* (the mode of the launch being made) */
    int side_by_side = 0;                /* kernels of the FIRs' stream and of the cascades' have been seen to run at once (probe_side_by_side): without that no ready words */
    std::vector<std::pair<hipStream_t, int>> probed;      /* ... per stream the FIRs were launched on */
    std::vector<hipStream_t> retired;                     /* cascades' streams that shared a hardware queue with a FIRs' stream (probe_side_by_side) */
    int remade = 0;                                       /* ... how many times the cascades' stream was made anew */
    /* cascade launch groups of one plan side by side (round 5: launch_cascades) */
    hipStream_t bq_side[3] = {nullptr, nullptr, nullptr};
    hipEvent_t bq_fork = nullptr, bq_join[3] = {nullptr, nullptr, nullptr};
    int group_fanout = 1;               /* "group_fanout" 0: the groups one after the other on one stream, as through round 4 */
    int cu_split = 0;                                     /* "cu_split" (experiment, round 5): > 0: the cascades' stream runs on that many CUs (a CU mask: the first cu_split / 8 CUs of
                                                             every XCD), the FIRs on a stream of the library's own with the complementary mask */
    unsigned *d_ready_timeouts = nullptr;        /* [0] waves whose bounded wait for a ready word ran out (never, see chain_ready_wait); [2..3] the device address of h_ready_flag */
    unsigned *h_ready_flag = nullptr;            /* mapped pinned host word such a wave sets: the host sees it without a copy or a synchronisation (ready_check) */
    int ready_test = 0;                          /* tests only: that many coming launches of "ready_words" 2 skip their ready_set kernel, so that their FIR waves time out */
    int *d_tag_prev = nullptr;           /* tagoutput: the plugin's `previoussample` */
    /* host-pointer block calls: the caller's buffers pinned in place (cache), copies and kernels on three streams */
    struct Pinned { const void *ptr; size_t bytes; bool ours; int refs; };   /* refs: queued blocks still copying from / into it */
    std::vector<Pinned> pinned;
    hipStream_t s_h2d = nullptr, s_run = nullptr, s_d2h = nullptr;
    std::vector<hipEvent_t> ev_host;
    /* queued host-pointer blocks (avdsp_hip_submit_block_host): a ring of staging pairs and three streams of their own --
     * created WITHOUT hipStreamNonBlocking, so that everything else the library does on the null stream (state reads,
     * resets, the synchronous block calls) is ordered behind the blocks in flight */
    static constexpr int kHostQueue = 4;
    struct HostSlot { unsigned *d_in = nullptr, *d_out = nullptr; size_t in_cap = 0, out_cap = 0; hipEvent_t h2d = nullptr, run = nullptr, d2h = nullptr;
                      const void *h_in = nullptr, *h_out = nullptr; };      /* the caller's buffers of the block in this slot (pin references) */
    HostSlot hq[kHostQueue];
    hipStream_t q_h2d = nullptr, q_run = nullptr, q_d2h = nullptr;
    unsigned long long hq_submitted = 0, hq_waited = 0;
    hipEvent_t input_ready = nullptr;    /* set around a run_block whose input block is still being written by another stream (a copy, pcm_unpack): the overlap mode's cascade waits for it */
    hipEvent_t ev_unpack = nullptr;      /* ... recorded behind the library's own pcm_unpack */
    int host_split = 0;                  /* frames per piece of a host block (0 = one piece; pieces only pay with host_pin) */
    int host_pin = 0;                    /* pin the caller's buffers in place and remember them: only for a host that keeps them allocated */
    bool ev_fir_set[kAhead] = {false, false, false};
    unsigned long long blk = 0;
};

namespace {

int pow2ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }

/* A FIR wave whose bounded wait for its cascade's ready word ran out (chain_ready_wait) has summed a window its cascade may not
 * have written: the block it belongs to is not the reference's.  The library has long returned 0 for that block (launches are
 * asynchronous), so the failure is STICKY: from the moment the wave's mark is visible every entry point that touches the program
 * fails with this text until the caller acknowledges it (dspRuntimeReset, or dspRuntimeSetOption("ready_timeouts", 0) after
 * re-uploading a state it trusts).  The reference's failures are return codes, never silent (dsp_runtime.c:150-195). */
int ready_check(avdsp_hip_prog *prog)
{
    if (!prog || !prog->h_ready_flag || !*(volatile unsigned *)prog->h_ready_flag) return 0;
    set_err("a FIR wave gave up waiting for its cascade's ready word (\"ready_words\" %d): at least one earlier block's output and the FIR "
            "state behind it are not valid; dspRuntimeReset() or dspRuntimeSetOption(\"ready_timeouts\", 0) acknowledges", prog->ready_mode_now ? prog->ready_mode_now : prog->ready_words);
    g_err_ready = true;
    return -1;
}
#define READY_CHECK(prog) do { if (ready_check(prog)) return -1; } while (0)

/* ---- copies between the CALLER's memory and the device: the GPU never touches pageable memory it does not own (round 5) ----
 * hipMemcpy[Async] of pageable memory: up to ~1 MB this runtime copies through a pinned buffer of its own; from 4 MB on it PINS THE CALLER'S
 * PAGES and lets the copy engine read / write them (tools/pageable_copy_probe.hip: the engine's addresses are the caller's), and such a pin
 * outlives the call -- it stays with the stream until that stream's next wait.  This library's input stream was never waited for.  A buffer
 * that is freed, whose heap pages are trimmed away and come back, and that is handed over again at the same address and size then meets the
 * old pin: "Memory access fault by GPU node-2 ... on address 0x5c87... (a heap address)", no wave active in the core dump (rocgdb: a copy
 * engine, not a kernel) -- three full test runs of five died that way this round, one of them inside PyTorch's own .cuda() of a 4 MB numpy array.
 * So: every copy of 1 MB or more between memory the caller owns and the device goes through two pinned chunks of the library's own (the CPU
 * copies a chunk while the engine moves the previous one), whatever the runtime would have done.  Memory the CALLER has pinned -- "host_pin",
 * the queued calls, hipHostRegister / hipHostMalloc of his own -- is told apart by hipPointerGetAttributes and copied directly. */
constexpr size_t kBounceFrom = 1u << 20, kBounceChunk = 4u << 20;
struct Bounce { char *h[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr}; hipStream_t s = nullptr; };
Bounce g_bounce[16];
std::mutex g_bounce_mutex;

bool caller_memory_is_pinned(const void *p)
{
    hipPointerAttribute_t at;
    const hipError_t e = hipPointerGetAttributes(&at, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }      /* (an address the runtime has never heard of: pageable) */
    return at.type == hipMemoryTypeHost || at.type == hipMemoryTypeManaged || at.type == hipMemoryTypeDevice;
}

int bounce_ready(Bounce **out)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    Bounce &b = g_bounce[dev & 15];
    if (!b.s) {
        HIP_TRY(hipStreamCreateWithFlags(&b.s, hipStreamNonBlocking));
        for (int k = 0; k < 2; k++) {
            HIP_TRY(hipHostMalloc((void **)&b.h[k], kBounceChunk, hipHostMallocDefault));
            HIP_TRY(hipEventCreateWithFlags(&b.ev[k], hipEventDisableTiming));
        }
    }
    *out = &b;
    return 0;
}

/* both like hipMemcpy: the device side is idle or ordered by the caller of these (they run on a stream of their own and wait for it) */
int copy_from_caller(void *d_dst, const void *h_src, size_t bytes)
{
    if (!bytes) return 0;
    if (bytes < kBounceFrom || caller_memory_is_pinned(h_src)) { HIP_TRY(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice)); return 0; }
    std::lock_guard<std::mutex> lock(g_bounce_mutex);
    Bounce *b;
    if (bounce_ready(&b)) return -1;
    size_t off = 0;
    for (int c = 0; off < bytes; c++, off += kBounceChunk) {
        const size_t n = std::min(kBounceChunk, bytes - off);
        if (c >= 2) HIP_TRY(hipEventSynchronize(b->ev[c & 1]));          /* the chunk this buffer held two turns ago has left it */
        memcpy(b->h[c & 1], (const char *)h_src + off, n);
        HIP_TRY(hipMemcpyAsync((char *)d_dst + off, b->h[c & 1], n, hipMemcpyHostToDevice, b->s));
        HIP_TRY(hipEventRecord(b->ev[c & 1], b->s));
    }
    HIP_TRY(hipStreamSynchronize(b->s));
    return 0;
}

int copy_to_caller(void *h_dst, const void *d_src, size_t bytes)
{
    if (!bytes) return 0;
    if (bytes < kBounceFrom || caller_memory_is_pinned(h_dst)) { HIP_TRY(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost)); return 0; }
    std::lock_guard<std::mutex> lock(g_bounce_mutex);
    Bounce *b;
    if (bounce_ready(&b)) return -1;
    const int nc = (int)((bytes + kBounceChunk - 1) / kBounceChunk);
    auto chunk_bytes = [&](int c) { return std::min(kBounceChunk, bytes - (size_t)c * kBounceChunk); };
    for (int c = 0; c <= nc; c++) {                       /* chunk c is on its way while the CPU takes chunk c - 1 home */
        if (c < nc) {                                     /* (its buffer was emptied by the CPU a turn ago) */
            HIP_TRY(hipMemcpyAsync(b->h[c & 1], (const char *)d_src + (size_t)c * kBounceChunk, chunk_bytes(c), hipMemcpyDeviceToHost, b->s));
            HIP_TRY(hipEventRecord(b->ev[c & 1], b->s));
        }
        if (c >= 1) {
            HIP_TRY(hipEventSynchronize(b->ev[(c - 1) & 1]));
            memcpy((char *)h_dst + (size_t)(c - 1) * kBounceChunk, b->h[(c - 1) & 1], chunk_bytes(c - 1));
        }
    }
    return 0;
}

template <typename T>
int upload_vec(T **dst, const std::vector<T> &v)
{
    *dst = nullptr;
    if (v.empty()) return 0;
    HIP_TRY(hipMalloc((void **)dst, v.size() * sizeof(T)));
    return copy_from_caller(*dst, v.data(), v.size() * sizeof(T));
}

void free_plan(Plan &p)
{
    (void)hipFree(p.d_chains); (void)hipFree(p.d_sec_coef); (void)hipFree(p.d_sec_state);
    for (auto &g : p.bq) {
        (void)hipFree(g.d_ids); (void)hipFree(g.d_rows); (void)hipFree(g.d_lanes); (void)hipFree(g.d_scratch[0]); (void)hipFree(g.d_scratch[1]);
        for (auto &pc : g.pieces) { (void)hipFree(pc.d_ids); (void)hipFree(pc.d_rows); (void)hipFree(pc.d_lanes); }
    }
    (void)hipFree(p.d_rows_all); (void)hipFree(p.d_lanes_all);
    (void)hipFree(p.d_sops); (void)hipFree(p.d_sargs); (void)hipFree(p.d_lseq); (void)hipFree(p.d_lane_rows);
    (void)hipFree(p.d_fir_ids); (void)hipFree(p.d_pass_ids); (void)hipFree(p.d_ring); (void)hipFree(p.d_ring64); (void)hipFree(p.d_own); (void)hipFree(p.d_taps64);
    (void)hipFree(p.d_ready);
}

int fir_groups_per_chunk(int max_taps)
{
    const int G = (max_taps + 15 + 15) >> 4;
    const int nc = (G + kMaxGpc - 1) / kMaxGpc;
    const int gpc = (G + nc - 1) / nc;
    return std::min((gpc + kNG - 1) / kNG * kNG, kMaxGpc);
}

size_t fir_lds_bytes(int gpc, int *hs_cap, int *row)
{
    *hs_cap = (fir_hs_len(gpc) + 3) & ~3;
    *row = fir_win_row(gpc);
    return (size_t)(*hs_cap + 16 * *row) * sizeof(double);
}

Ring plan_ring(const Plan &pl) { return Ring{pl.d_ring, pl.ring_R, pl.wpos, pl.d_ring64}; }

hipEvent_t take_event(avdsp_hip_prog *prog)
{
    if (!prog->free_events.empty()) { hipEvent_t e = prog->free_events.back(); prog->free_events.pop_back(); return e; }
    /* timers order nothing and hand no data to the host: no system-scope release behind the kernel they ride on (timing stays on) */
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) return nullptr;
    return e;
}

/* Kernel timers (dspRuntimeKernelTime).  A launch that goes through hipExtLaunchKernel carries its own pair of events: they
 * take the dispatch's own start and end stamps -- what rocprofv3's kernel trace reads -- and put nothing on the stream (ride()).
 * Other launches are bracketed by two recorded events (begin() .. destructor), which costs the stream ~5 us per pair and reads
 * that much more than the kernel took. */
struct ProfileScope {
    avdsp_hip_prog *prog; hipStream_t stream; int kind; bool on = false; hipEvent_t a = nullptr;
    ProfileScope(avdsp_hip_prog *p, hipStream_t s, int k) : prog(p), stream(s), kind(k)
    {
        on = (prog->profile >> kind & 1u) && prog->profile_seen[kind & 7]++ % (unsigned)prog->profile_stride == 0;
    }
    void keep(hipEvent_t s, hipEvent_t e, bool pair)
    {
        if (prog->spans.size() >= 65536) {                 /* nobody reads the timers: forget the oldest half (their events go back to the pool) */
            for (size_t i = 0; i < 32768; i++) {
                for (int k = 0; k < avdsp_hip_prog::kAhead; k++)
                    if (prog->ev_fir_now[k] == prog->spans[i].b) { (void)hipEventSynchronize(prog->spans[i].b); prog->ev_fir_now[k] = prog->ev_fir[k]; prog->ev_fir_set[k] = false; }
                prog->free_events.push_back(prog->spans[i].a); prog->free_events.push_back(prog->spans[i].b);
            }
            prog->spans.erase(prog->spans.begin(), prog->spans.begin() + 32768);
        }
        prog->spans.push_back({kind, s, e, pair});
    }
    bool ride(hipEvent_t &s, hipEvent_t &e)                /* events for hipExtLaunchKernel's start / stop slots, if this launch is sampled */
    {
        if (!on) return false;
        s = take_event(prog); e = take_event(prog);
        if (!s || !e) { if (s) prog->free_events.push_back(s); if (e) prog->free_events.push_back(e); return false; }
        keep(s, e, false);
        prog->last_ride_stop = e;
        return true;
    }
    void begin() { if (on && !a && (a = take_event(prog))) (void)hipEventRecord(a, stream); }
    ~ProfileScope()
    {
        if (!a) return;
        hipEvent_t b = take_event(prog);
        if (!b) { prog->free_events.push_back(a); return; }
        (void)hipEventRecord(b, stream);
        keep(a, b, true);
    }
};

/* a launch that is timed, when it is, by the dispatch's own stamps */
template <typename Args>
int launch_timed(ProfileScope &scope, const void *fn, dim3 grid, dim3 block, size_t lds, hipStream_t stream, Args &a, hipEvent_t stop = nullptr)
{
    void *kargs[] = {(void *)&a};
    hipEvent_t s = nullptr, e = nullptr;
    if (stop) { scope.begin(); HIP_TRY(hipExtLaunchKernel(fn, grid, block, kargs, lds, stream, nullptr, stop, 0)); return 0; }
    if (scope.ride(s, e)) { HIP_TRY(hipExtLaunchKernel(fn, grid, block, kargs, lds, stream, s, e, 0)); return 0; }
    if (scope.kind == AVDSP_KERNEL_FIR && scope.prog->fir_mode_now == 2) {
        avdsp_hip_prog *pg = scope.prog;
        const unsigned k = (pg->launch_ev_next++ & 7u) * 2;
        for (unsigned j = k; j < k + 2; j++) if (!pg->launch_ev[j]) HIP_TRY(hipEventCreate(&pg->launch_ev[j]));
        HIP_TRY(hipExtLaunchKernel(fn, grid, block, kargs, lds, stream, pg->launch_ev[k], pg->launch_ev[k + 1], 0));
        return 0;
    }
    HIP_TRY(hipLaunchKernel(fn, grid, block, kargs, lds, stream));
    return 0;
}

#ifdef AVDSP_BQ_STAMPS
static unsigned long long *g_bq_stamps = nullptr; static int g_bq_stamp_waves = 0;
extern "C" int avdsp_hip_debug_bq_stamps(unsigned long long *host_out, int max_waves)
{
    if (!g_bq_stamps) return 0;
    const int n = g_bq_stamp_waves < max_waves ? g_bq_stamp_waves : max_waves;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpy(host_out, g_bq_stamps, (size_t)n * 32 * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return n;
}
#endif

template <int FMT>
int launch_biquad(avdsp_hip_prog *prog, Plan &pl, const Plan::Group &g, const int *ids, int n, BlockIO io,
                  int biquad_impl, hipStream_t stream, hipEvent_t stop = nullptr, bool with_ready = false)
{
    ProfileScope scope(prog, stream, AVDSP_KERNEL_BIQUAD);
    BiquadArgs a{};
    a.buf = prog->d_buf; a.chains = pl.d_chains; a.sec_coef = pl.d_sec_coef; a.sec_state = pl.d_sec_state;
    a.group = ids; a.ngroup = n; a.nsec = g.nsec; a.ring = plan_ring(pl); a.io = io;
    a.ready = with_ready ? pl.d_ready : nullptr; a.seq = pl.seq;
      /* (a launch whose FIR waits for the words; its ring stores are then write-through) */
#ifdef AVDSP_BQ_STAMPS
    {
        static unsigned long long *d_stamps = nullptr;
        if (!d_stamps) { HIP_TRY(hipMalloc((void **)&d_stamps, (size_t)65536 * 4 * 32 * 8)); }
        HIP_TRY(hipMemsetAsync(d_stamps, 0, (size_t)65536 * 4 * 32 * 8, stream));
        a.stamps = d_stamps; g_bq_stamps = d_stamps; g_bq_stamp_waves = ((n * g.P + kBlock - 1) / kBlock + 7) / 8 * 8 * 4;
    }
#endif
    if (biquad_impl == 0 || g.P > 64) {
        scope.begin();
        hipLaunchKernelGGL(biquad_simple<FMT>, dim3((n + 63) / 64), dim3(64), 0, stream, a);
        if (stop) HIP_TRY(hipEventRecord(stop, stream));
    } else if (biquad_impl == 1 && g.P == 16 && g.d_rows) {
        /* one 16-lane row per chain: biquad_row (format 4 only where the cascade feeds a FIR -- a format-4 STORE needs all of the
         * accumulator, biquad_row hands on its float), biquad_row_i64 */
        if constexpr (FMT == 2) {
            const int nblk = (n + 15) / 16;
            a.rows = g.d_rows; a.lanes = g.d_lanes;
            a.per_xcd = (nblk + 7) / 8;
            if (launch_timed(scope, (const void *)biquad_row_i64, dim3(a.per_xcd * 8), dim3(kBlock), 0, stream, a, stop)) return -1;
        } else {
            const int nblk = (n + 15) / 16;
            a.rows = g.d_rows; a.lanes = g.d_lanes;
            a.per_xcd = (nblk + 7) / 8;
            /* (under "overlap" `stop` rides on the kernel's own completion signal instead of a marker packet behind it: 7.1 instead of
             * 8.3 us to the start of the kernel that waits for it on another stream, tools/stream_handover_bench.hip) */
            /* (format 4: rows that store int samples need the accumulator itself -- the ACC form; FIR-feeding chains and pieces hand a float on) */
            const void *fn = (const void *)biquad_row<FMT>;
            if constexpr (FMT == 4) { if (!(g.all_fir || g.raw_out)) fn = (const void *)biquad_row<4, true>; }
            if (launch_timed(scope, fn, dim3(a.per_xcd * 8), dim3(kBlock), 0, stream, a, stop)) return -1;
        }
    } else {
        const int cpb = kBlock / g.P;
        const int nblk = (n + cpb - 1) / cpb;
        a.per_xcd = (nblk + 7) / 8;
        const void *fn = g.P == 1 ? (const void *)biquad_pipe<FMT, 1> : g.P == 2 ? (const void *)biquad_pipe<FMT, 2> : g.P == 4 ? (const void *)biquad_pipe<FMT, 4>
                       : g.P == 8 ? (const void *)biquad_pipe<FMT, 8> : g.P == 16 ? (const void *)biquad_pipe<FMT, 16> : g.P == 32 ? (const void *)biquad_pipe<FMT, 32>
                       : (const void *)biquad_pipe<FMT, 64>;
        if (launch_timed(scope, fn, dim3(a.per_xcd * 8), dim3(kBlock), 0, stream, a, stop)) return -1;
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

#ifdef AVDSP_FIR_STAMPS
static unsigned long long *g_fir_stamps = nullptr; static int g_fir_stamp_waves = 0;
extern "C" int avdsp_hip_debug_fir_stamps(unsigned long long *host_out, int max_waves)
{
    if (!g_fir_stamps) return 0;
    const int n = std::min(max_waves, g_fir_stamp_waves);
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpy(host_out, g_fir_stamps, (size_t)n * 32 * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return n;
}
#endif

template <int FMT, int R, bool BIG = false, bool SPLIT = false>
int launch_fir_tile(avdsp_hip_prog *prog, Plan &pl, const int *ids, int n, BlockIO io, hipStream_t stream, ProfileScope &scope, bool wait_ready, hipEvent_t stop = nullptr, bool lean = true)
{
    FirTileArgs a{};
    a.buf = prog->d_buf; a.chains = pl.d_chains; a.group = ids; a.ngroup = n;
    a.ring = plan_ring(pl); a.io = io; a.taps64 = pl.d_taps64; a.pitch64 = pl.pitch64;
    if (wait_ready) { a.ready = pl.d_ready; a.seq = pl.seq; a.timeouts = prog->d_ready_timeouts; a.ready_acquire = prog->ready_mode_now != 2 || prog->overlap >= 2; }
    /* the tiles a chain has in this launch (a power of two, so that a chain's waves sit in one workgroup): a short block is fewer tiles
     * than the TileGeom's 1024 frames, and waves without a tile would only hold their workgroup's LDS */
    constexpr int kWpc = TileGeom<R, BIG>::WPC;
    const int tiles = (io.nframes + TileGeom<R, BIG>::FW - 1) / TileGeom<R, BIG>::FW;
    /* (BIG -- at most a wave per SIMD -- keeps the four-quarters arrangement: there a wave that leaves at once frees nothing anybody waits
     * for, and its workgroup's other waves sit on CUs of their own: 256 chains x 4096 taps at 64 .. 512 frames 36.3-37.6 us against
     * 38.1-38.8 regrouped) */
    /* ... as long as its workgroups -- one per chain, one per CU: 140 KB of LDS -- fit the chip at once.  512 or 1024 chains at 256 frames
     * are 512 / 1024 workgroups of ONE live wave each, two or four rounds of 36 us (a 512-chain shard's 256-frame block took 91 us,
     * longer than its 1024-frame block): those regroup like everybody else. */
    const bool quarters = SPLIT || (BIG && n <= (prog->num_cus > 0 ? prog->num_cus : 256));
    const int wpc = quarters ? kWpc : tiles <= 1 ? 1 : tiles <= 2 ? std::min(2, kWpc) : kWpc;
    a.wpc_shift = wpc == 4 ? 2 : wpc == 2 ? 1 : 0;
    const int nwg = (n * wpc * (SPLIT ? 2 : 1) + 3) / 4;
    a.per_xcd = (nwg + 7) / 8;
    const size_t lds = (size_t)4 * TileGeom<R, BIG>::LDS_DOUBLES * sizeof(double) + 64;      /* + the four words the waves exchange at the end */
#ifdef AVDSP_FIR_STAMPS
    static unsigned long long *d_stamps = nullptr;
    if (!d_stamps) { HIP_TRY(hipMalloc((void **)&d_stamps, (size_t)8192 * 4 * 32 * 8)); }
    HIP_TRY(hipMemsetAsync(d_stamps, 0, (size_t)8192 * 4 * 32 * 8, stream));
    a.stamps = d_stamps;
    g_fir_stamps = d_stamps; g_fir_stamp_waves = a.per_xcd * 8 * 4;
#endif
    return launch_timed(scope, lean ? (const void *)fir_tile<FMT, R, BIG, SPLIT, true> : (const void *)fir_tile<FMT, R, BIG, SPLIT, false>,
                        dim3(a.per_xcd * 8), dim3(kBlock), lds, stream, a, stop);
}


template <int FMT, int R>
int launch_fir_stream(avdsp_hip_prog *prog, Plan &pl, const int *ids, int n, BlockIO io, hipStream_t stream)
{
    if (!pl.d_ring64) {
        /* first use: the operand ring (8 bytes per ring entry) is made from the float ring; every ring writer keeps it up from here on */
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMalloc((void **)&pl.d_ring64, (size_t)pl.nchains * 2 * pl.ring_R * sizeof(double)));      /* (every operand twice: Ring) */
        RingConvArgs ca{prog->d_buf, pl.d_chains, pl.d_fir_ids, pl.n_fir, plan_ring(pl)};
        hipLaunchKernelGGL(ring_widen, dim3(pl.n_fir), dim3(kBlock), 0, stream, ca);
        HIP_TRY(hipGetLastError());
    }
    FirTileArgs a{};
    a.buf = prog->d_buf; a.chains = pl.d_chains; a.group = ids; a.ngroup = n;
    a.ring = plan_ring(pl); a.io = io; a.taps64 = pl.d_taps64; a.pitch64 = pl.pitch64;
    if (pl.n_fir_only) {
        const long long total = (long long)n * io.nframes;
        hipLaunchKernelGGL(fir_feed<FMT>, dim3((unsigned)std::min<long long>((total + kBlock - 1) / kBlock, 4096)), dim3(kBlock), 0, stream, a);
        HIP_TRY(hipGetLastError());
    }
    /* one wave per SIMD at most: 256 workgroups of four waves, each wave takes its units in turn */
    const int tiles = (io.nframes + StreamGeom<R>::FW - 1) / StreamGeom<R>::FW;
    const int nwg = std::min((n * tiles + 3) / 4, prog->num_cus > 0 ? prog->num_cus : 256);
    a.per_xcd = (nwg + 7) / 8;
    const size_t lds = (size_t)4 * StreamGeom<R>::LDS_DOUBLES * sizeof(double) + 64;
#ifdef AVDSP_FIR_STAMPS
    static unsigned long long *d_stamps = nullptr;
    if (!d_stamps) { HIP_TRY(hipMalloc((void **)&d_stamps, (size_t)8192 * 4 * 32 * 8)); }
    HIP_TRY(hipMemsetAsync(d_stamps, 0, (size_t)8192 * 4 * 32 * 8, stream));
    a.stamps = d_stamps;
    g_fir_stamps = d_stamps; g_fir_stamp_waves = a.per_xcd * 8 * 4;
#endif
    hipLaunchKernelGGL((fir_stream<FMT, R>), dim3(a.per_xcd * 8), dim3(kStreamBlock), lds, stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int FMT, int R, bool BIG = false>
int launch_fir_flow(avdsp_hip_prog *prog, Plan &pl, const int *ids, int n, BlockIO io, hipStream_t stream, ProfileScope &scope, bool wait_ready, hipEvent_t stop)
{
    if (!pl.d_ring64) {
        /* first use: the operand ring is made from the float ring; every ring writer keeps it up from here on */
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMalloc((void **)&pl.d_ring64, (size_t)pl.nchains * 2 * pl.ring_R * sizeof(double)));
        RingConvArgs ca{prog->d_buf, pl.d_chains, pl.d_fir_ids, pl.n_fir, plan_ring(pl)};
        hipLaunchKernelGGL(ring_widen, dim3(pl.n_fir), dim3(kBlock), 0, stream, ca);
        HIP_TRY(hipGetLastError());
    }
    FirTileArgs a{};
    a.buf = prog->d_buf; a.chains = pl.d_chains; a.group = ids; a.ngroup = n;
    a.ring = plan_ring(pl); a.io = io; a.taps64 = pl.d_taps64; a.pitch64 = pl.pitch64;
    if (wait_ready) { a.ready = pl.d_ready; a.seq = pl.seq; a.timeouts = prog->d_ready_timeouts; a.ready_acquire = prog->ready_mode_now != 2 || prog->overlap >= 2; }
    a.wpc_shift = FlowGeom<R, BIG>::WPC == 4 ? 2 : FlowGeom<R, BIG>::WPC == 2 ? 1 : 0;      /* (fir_flow keeps the 1024-frame arrangement) */
    const int nwg = (n * FlowGeom<R, BIG>::WPC + 3) / 4;
    a.per_xcd = (nwg + 7) / 8;
    const size_t lds = (size_t)4 * FlowGeom<R, BIG>::LDS_DOUBLES * sizeof(double) + 64;
#ifdef AVDSP_FIR_STAMPS
    static unsigned long long *d_stamps = nullptr;
    if (!d_stamps) { HIP_TRY(hipMalloc((void **)&d_stamps, (size_t)8192 * 4 * 32 * 8)); }
    HIP_TRY(hipMemsetAsync(d_stamps, 0, (size_t)8192 * 4 * 32 * 8, stream));
    a.stamps = d_stamps;
    g_fir_stamps = d_stamps; g_fir_stamp_waves = a.per_xcd * 8 * 4;
#endif
    return launch_timed(scope, (const void *)fir_flow<FMT, R, BIG>, dim3(a.per_xcd * 8), dim3(kBlock), lds, stream, a, stop);
}

/* fir_impl: 0 = fir_plain (the reference's loop), 1 = fir_tile (default), 2 = fir_mfma (round 1's workgroup-per-channel kernel),
 * 3 = fir_stream, 4 = fir_flow */
template <int FMT>
int launch_fir(avdsp_hip_prog *prog, Plan &pl, const int *ids, int n, BlockIO io, int fir_impl, hipStream_t stream, bool wait_ready = false, hipEvent_t stop = nullptr)
{
    if constexpr (FMT == 2) { (void)prog; (void)pl; (void)ids; (void)n; (void)io; (void)fir_impl; (void)stream; (void)wait_ready; (void)stop; return 0; }
    else {
        ProfileScope scope(prog, stream, AVDSP_KERNEL_FIR);
        if (fir_impl == 4) {
            int rows = prog->fir_rows;
            if (rows != 1 && rows != 2 && rows != 4) rows = n >= 2048 ? 4 : n >= 1024 ? 2 : 1;
            while (rows > 1 && 128 * rows >= io.nframes) rows >>= 1;
            if (rows == 1 && (long long)n * ((io.nframes + 255) / 256) <= 1024 && prog->fir_rows != 1)     /* at most a wave per SIMD: long chunks, two window images */
                return launch_fir_flow<FMT, 1, true>(prog, pl, ids, n, io, stream, scope, wait_ready, stop);
            return rows == 4 ? launch_fir_flow<FMT, 4>(prog, pl, ids, n, io, stream, scope, wait_ready, stop)
                 : rows == 2 ? launch_fir_flow<FMT, 2>(prog, pl, ids, n, io, stream, scope, wait_ready, stop)
                             : launch_fir_flow<FMT, 1>(prog, pl, ids, n, io, stream, scope, wait_ready, stop);
        }
        if (fir_impl != 1) scope.begin();
        if (fir_impl == 3) {
            /* row tiles per wave: as many as leave every SIMD a wave (1024) */
            int rows = prog->fir_rows;
            const int tiles1 = (io.nframes + 255) / 256;
            if (rows != 1 && rows != 2 && rows != 4) rows = (long long)n * tiles1 >= 4 * 1024 ? 4 : (long long)n * tiles1 >= 2 * 1024 ? 2 : 1;
            while (rows > 1 && 128 * rows >= io.nframes) rows >>= 1;
            return rows == 4 ? launch_fir_stream<FMT, 4>(prog, pl, ids, n, io, stream)
                 : rows == 2 ? launch_fir_stream<FMT, 2>(prog, pl, ids, n, io, stream)
                             : launch_fir_stream<FMT, 1>(prog, pl, ids, n, io, stream);
        }
        if (fir_impl == 1) {
            /* row tiles per wave: as many as leave the chip two waves per SIMD (2048) -- a bigger tile reads fewer operands per MFMA */
            int rows = prog->fir_rows;
            if (rows != 1 && rows != 2 && rows != 4) {
                /* Round 5 (tools/regime_scan.sh): the chip holds 2048 of these waves at a time, a wave of R row tiles lasts R units, and a launch
                 * is over when its LAST round of waves is -- 3000 chains at four row tiles were 1.46 rounds, i.e. two: 480 us, as long as 4096
                 * chains.  So: the R with the fewest units, rounds(R) x R / efficiency(R) (0.90 / 0.84 / 0.79 of the matrix pipe at 4 / 2 / 1 row
                 * tiles).  For 512 / 1024 / 2048 / 4096 / 16384 chains that is what the thresholds chose (1 / 2 / 4 / 4 / 4). */
                double best = 1e30;
                rows = 1;
                for (int r : {4, 2, 1}) {
                    if (r > 1 && 128 * r >= io.nframes) continue;           /* a tile twice the block would multiply zeros */
                    const long long waves = (long long)n * ((io.nframes + 256 * r - 1) / (256 * r));
                    const double cost = (double)((waves + 2047) / 2048) * r / (r == 4 ? 0.90 : r == 2 ? 0.84 : 0.79);
                    if (cost < best - 1e-9) { best = cost; rows = r; }
                }
            }
            while (rows > 1 && 128 * rows >= io.nframes) rows >>= 1;       /* a tile twice the block would multiply zeros */
            /* one row tile and at most a wave per SIMD (1024): chunks twice as long -- nothing hides a boundary there */
            const long long waves1 = (long long)n * ((io.nframes + 255) / 256);
            /* "fir_split" (opt-in, not the reference's summation order): such a launch with two waves per tile instead */
            /* the lean chunk boundary (fir_tile, LEAN) where it was measured to win (tools/fir_boundary_lab.sh, one box, long / lean):
             * plans without cascades in front (256 chains x 4096 taps: 40.9 -> 39.4 us per step) and launches of more than one round of
             * waves (4096 chains: 0.511 -> 0.506 ms).  In between, the next blocks' cascades run beside the FIR and live on the long
             * boundary's bubbles: 2048 chains 0.259 -> 0.269 ms, 1024 chains 0.147 -> 0.154, 512 chains 0.0876 -> 0.0966. */
            const bool lean = prog->fir_lean >= 0 ? prog->fir_lean != 0 : (pl.bq.empty() || (long long)pl.n_fir * pl.max_taps >= 12000000ll);
            if (rows == 1 && waves1 <= 1024 && prog->fir_split) return launch_fir_tile<FMT, 1, false, true>(prog, pl, ids, n, io, stream, scope, wait_ready, stop, lean);
            if (rows == 1 && waves1 <= 1024 && prog->fir_rows != 1) return launch_fir_tile<FMT, 1, true>(prog, pl, ids, n, io, stream, scope, wait_ready, stop, lean);
            return rows == 4 ? launch_fir_tile<FMT, 4>(prog, pl, ids, n, io, stream, scope, wait_ready, stop, lean)
                 : rows == 2 ? launch_fir_tile<FMT, 2>(prog, pl, ids, n, io, stream, scope, wait_ready, stop, lean)
                             : launch_fir_tile<FMT, 1>(prog, pl, ids, n, io, stream, scope, wait_ready, stop, lean);
        }
        FirArgs a{};
        a.buf = prog->d_buf; a.chains = pl.d_chains; a.group = ids; a.ngroup = n;
        a.ring = plan_ring(pl); a.io = io;
        a.per_xcd = (n + 7) / 8;
        const int nwaves = 4;                            /* always 4: idle tiles' waves still stage */
        a.gpc = pl.fir_gpc;
        size_t lds = 0;
        if (fir_impl) lds = fir_lds_bytes(a.gpc, &a.hs_cap, &a.win_row);
        /* LDS opt-in was done at plan creation.  Few workgroups per CU: the deeper operand sets */
        auto kern = !fir_impl ? fir_plain<FMT> : (n <= 2 * 256 ? fir_mfma<FMT, 2> : fir_mfma<FMT, 1>);
        hipLaunchKernelGGL(kern, dim3(a.per_xcd * 8), dim3(64 * nwaves), lds, stream, a);
        HIP_TRY(hipGetLastError());
        return 0;
    }
}

/* The cascade and the FIR of a block normally run back to back on the caller's stream.
 *
 * "overlap" (opt-in, dspRuntimeSetOption): the cascade of block k+1 runs under the FIR of block k, on a stream of its
 * own in its ordinary launch geometry.  The cascade is a latency-bound recurrence -- 8 us + 40 ns per frame whatever the
 * channel count, a wave issuing 13 instructions per ~100 cycles -- and fir_tile's waves are independent of each other, so
 * the two share SIMDs well: a SIMD that hosts a cascade wave runs its FIR waves slower for the cascade's duration, the
 * others not at all, and the step shortens by 5 % (4096 channels on one GPU) to 8 % (a 512-channel shard).  (Round 2
 * first tried cascade workgroups of 1024 threads that claimed a CU's LDS to keep FIR waves off their CU: sixteen cascade
 * waves on four SIMDs slow each other 2.5x, and the step got LONGER, 123 -> 147 us on the 512-channel shard.)  Ordering:
 *     cascade k   waits for FIR k-3 (the rings hold three launches of frames beside the longest history: the positions it appends
 *                 are free once that FIR has read its window).  Not k-2: a cascade wave that shares its SIMD with the FIR's MFMA
 *                 stream is served every few hundred cycles and ends about when that FIR does; the FIR that needs its output
 *                 would wait for it and for the hand-over between the two queues (7-11 us).  One block further ahead the
 *                 cascade runs under FIR k-2 and is long done when FIR k-1 ends: FIR k follows FIR k-1 in stream order.
 *                 NOT for the caller's stream -- the mode's contract is that the input block is complete in memory
 *                 when the call is made (stream order would put it behind FIR k-1, which is the very thing to avoid);
 *     FIR k       on the caller's stream, after cascade k: outputs are ordered on that stream as always, and so is
 *                 everything the caller enqueues after the call (it may overwrite the input block).
 * Only when every cascade feeds a FIR (a cascade that stores straight to the output block would write it from the
 * side stream).                                                                                                */
/* "ready_words" 2: the words of a launch's FIR chains set by a kernel of its own BEHIND the cascade on the cascades' stream -- the
 * kernel boundary in front of it is the release (the cascade's ring stores are written back and visible), so the cascade pays
 * nothing (mode 1: write-through stores and a drain per wave), and the FIR's stream carries no wait packet: a FIR follows the
 * previous one like any kernel of a queue, its waves look at their chain's word (long set: the cascades run a block ahead). */
__global__ void ready_set(unsigned *ready, const int *ids, int n, unsigned seq)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) ready[ids[i]] = seq;
}

/* Ready words need kernels of two queues to RUN side by side: the FIR's waves sit on their SIMDs until a kernel on the cascades' queue
 * has set their words.  Where dispatches are serialised -- `rocprofv3 --pmc` does that (round 4's PMC pass of the headline kernel
 * counted seconds of polling waves: every FIR ran into its bound), a debugger, a queue-per-process time slice -- the FIR would wait
 * for a kernel that cannot start.  So the mode is only taken after this has been SEEN to work, once per program: a kernel that
 * waits (bounded, ~3 ms) for a word, a kernel on the other queue that sets it.  Not seen: the FIRs wait for the cascades' events,
 * as with "ready_words" 0 -- slower by the wait packet, correct anywhere.  (A GPU that stops running two queues at once LATER still
 * ends in chain_ready_wait's bound, and that is an error at the C ABI: ready_check.) */
__global__ void probe_wait(const unsigned *flag, unsigned *seen)
{
    unsigned spins = 0, v = 0;
    while ((v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0 && ++spins < 4096) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0) __hip_atomic_store(seen, v ? 1u : 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void probe_set(unsigned *flag) { if (threadIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

/* the cascades' stream and the FIRs' own streams; with "cu_split" k: on k CUs / on the others (tools/cu_mask_probe.hip: mask bit i is
 * CU i / 8 of XCD i % 8, so the low k bits are the first k / 8 CUs of every XCD) */
static int make_side_stream(avdsp_hip_prog *prog, hipStream_t *st, bool cascades)
{
    const int ncu = prog->cu_split < 0 ? -prog->cu_split : prog->cu_split;      /* (< 0: only the cascades' stream is masked, the FIRs stay where they are) */
    if (ncu == 0 || (prog->cu_split < 0 && !cascades)) { HIP_TRY(hipStreamCreateWithFlags(st, hipStreamNonBlocking)); return 0; }
    uint32_t m[8];
    for (int w = 0; w < 8; w++) {
        uint32_t low = 0;
        for (int bit = 0; bit < 32; bit++) if (32 * w + bit < ncu) low |= 1u << bit;
        m[w] = cascades ? low : ~low;
    }
    HIP_TRY(hipExtStreamCreateWithCUMask(st, 8, m));
    return 0;
}

static int probe_once(avdsp_hip_prog *prog, hipStream_t fir_stream, bool *ok)
{
    *ok = false;
    if (!prog->d_ready_timeouts || !prog->h_ready_flag) return 0;
    unsigned *dseen = nullptr;
    HIP_TRY(hipHostGetDevicePointer((void **)&dseen, prog->h_ready_flag, 0));
    *(volatile unsigned *)(prog->h_ready_flag + 1) = 0;
    HIP_TRY(hipMemsetAsync(prog->d_ready_timeouts + 1, 0, 4, fir_stream));
    HIP_TRY(hipStreamSynchronize(fir_stream));
    HIP_TRY(hipStreamSynchronize(prog->s_bq));
    hipLaunchKernelGGL(probe_wait, dim3(1), dim3(64), 0, fir_stream, prog->d_ready_timeouts + 1, dseen + 1);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(probe_set, dim3(1), dim3(64), 0, prog->s_bq, prog->d_ready_timeouts + 1);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(fir_stream));
    HIP_TRY(hipStreamSynchronize(prog->s_bq));
    *ok = *(volatile unsigned *)(prog->h_ready_flag + 1) == 1u;
    return 0;
}

/* Per stream the FIRs are launched on: do its kernels and the cascades' stream's run side by side?  That depends on the hardware
 * queues the runtime gave the two streams -- it has a handful per device and deals them out in turn, so a caller's stream may well
 * share one with the cascades' stream.  Then the whole overlap mode is a fiction: cascade and FIR take turns on one queue and the
 * step is their SUM (round 5 saw it in the round's own bench lines: a 512-chain shard at 107.7 us instead of 88, a 1024-chain one
 * at 163.9 instead of 147 -- FIR + cascade to the microsecond -- in one process out of several; round 4's "two states of a shard"
 * were probably this as well), and a FIR that looks at ready words would wait for a kernel queued BEHIND it.  So the pair is
 * tried once per stream (a kernel that waits for a word, a kernel on the other stream that sets it), and when they do not run at
 * once the cascades' stream is made anew -- the next one the runtime hands out sits on another queue -- up to six times.  What is
 * left after that (dispatches serialised from outside: a counter-collecting profiler) runs on events, correct anywhere. */
static int probe_side_by_side(avdsp_hip_prog *prog, hipStream_t fir_stream)
{
    for (auto &pr : prog->probed) if (pr.first == fir_stream) { prog->side_by_side = pr.second; return 0; }
    prog->side_by_side = 0;
    if (prog->probed.size() >= 16) prog->probed.clear();
    bool ok = false;
    for (int attempt = 0; attempt < 6; attempt++) {
        if (probe_once(prog, fir_stream, &ok)) return -1;
        if (ok || !prog->d_ready_timeouts || prog->remade >= 8) break;
        /* the pair shares a queue (or nothing runs side by side here at all): another stream for the cascades.  The old one is kept
         * until the program goes (destroying it would hand its queue slot straight back), and every pair tried so far is void. */
        hipStream_t ns = nullptr;
        if (make_side_stream(prog, &ns, true)) return -1;
        HIP_TRY(hipStreamSynchronize(prog->s_bq));
        prog->retired.push_back(prog->s_bq);
        prog->s_bq = ns;
        prog->probed.clear();
        prog->remade++;
    }
    prog->side_by_side = ok;
    prog->probed.push_back({fir_stream, ok ? 1 : 0});
    return 0;
}

static int overlap_ready(avdsp_hip_prog *prog)
{
    if (prog->s_bq) return 0;
    if (make_side_stream(prog, &prog->s_bq, true)) return -1;
    for (auto &fs : prog->s_fir) if (make_side_stream(prog, &fs, false)) return -1;
    for (int i = 0; i < avdsp_hip_prog::kAhead; i++) {
        /* they order kernels of this device among themselves: no system-scope fence (tools/stream_handover_bench.hip: 8.1 instead of 10.6 us) */
        HIP_TRY(hipEventCreateWithFlags(&prog->ev_bq[i], hipEventDisableTiming | hipEventDisableSystemFence));
        HIP_TRY(hipEventCreateWithFlags(&prog->ev_fir[i], hipEventDisableTiming | hipEventDisableSystemFence));
    }
    return 0;
}

/* The cascades of a plan: one launch per section count (a row of biquad_row, a lane group of biquad_pipe, has ONE length).  A program of
 * equal chains -- every BASELINE config -- is one launch; a real crossover, two biquads on one way and six on another, is several, each a
 * latency-bound recurrence of ~32 us whatever it holds: 1024 chains with 1 .. 8 sections took 248 us on one stream, eight launches in a row
 * (tools/mixed_groups_bench.py).  The groups are independent chains: they go out over up to four streams (`st` and three of the library's
 * own, as many as the runtime has hardware queues), forked from `st` by an event and joined back into it, so that whatever follows on
 * `st` -- the block's FIR, the caller's next kernel -- is ordered behind all of them.  `last_stop`: an event the caller wants on the
 * cascades' completion (the overlap mode's ev_bq): recorded behind the join. */
template <int FMT>
int launch_cascades(avdsp_hip_prog *prog, Plan &pl, BlockIO io, int biquad_impl, hipStream_t st, hipEvent_t last_stop, bool with_ready)
{
    /* (i) every cascade of up to 16 sections in ONE biquad_row launch, whatever their lengths (the table of all rows, Plan::d_rows_all):
     * 1024 chains with 1 .. 8 sections 248 -> 33 us.  (ii) what that does not cover -- 17 sections and more, or the options that take
     * biquad_row out -- as before, one launch per length, side by side over the streams. */
    const bool merged = prog->group_fanout && biquad_impl == 1 && pl.d_rows_all;
    /* one group on one stream: a launch -- or, for cascades of more than 64 sections, the group's pieces one after the other, the words
     * between them through the group's scratch columns */
    auto launch_group = [&](const Plan::Group &g, hipStream_t s, hipEvent_t stop) -> int {
        if (g.pieces.empty()) return launch_biquad<FMT>(prog, pl, g, g.d_ids, g.n, io, biquad_impl, s, stop, with_ready);
        const size_t np = g.pieces.size();
        for (size_t k = 0; k < np; k++) {
            BlockIO pio = io;
            if (k > 0)      { pio.in = g.d_scratch[(k - 1) & 1]; pio.in_stride = g.n; pio.in_base = 0; }
            if (k + 1 < np) { pio.out = g.d_scratch[k & 1];      pio.out_stride = g.n; pio.out_base = 0; }
            const auto &pc = g.pieces[k];
            if (launch_biquad<FMT>(prog, pl, pc, pc.d_ids, pc.n, pio, biquad_impl, s, k + 1 == np ? stop : nullptr, k + 1 == np && with_ready)) return -1;
        }
        return 0;
    };
    std::vector<const Plan::Group *> todo;
    for (auto &g : pl.bq) if (!(merged && g.P == 16 && g.d_rows)) todo.push_back(&g);
    if (merged) {
        const Plan::Group m{16, 0, pl.n_rows_all, nullptr, pl.rows_all_fir, pl.d_rows_all, pl.d_lanes_all};
        if (launch_biquad<FMT>(prog, pl, m, nullptr, m.n, io, biquad_impl, st, todo.empty() ? last_stop : nullptr, with_ready)) return -1;
        if (todo.empty()) return 0;
    }
    const size_t ng = todo.size();
    if (ng < 2 || !prog->group_fanout) {
        for (size_t gi = 0; gi < ng; gi++) {                  /* (the last group's kernel carries the event: the stream is in order) */
            auto &g = *todo[gi];
            if (launch_group(g, st, gi + 1 == ng ? last_stop : nullptr)) return -1;
        }
        return 0;
    }
    if (!prog->bq_fork) {
        HIP_TRY(hipEventCreateWithFlags(&prog->bq_fork, hipEventDisableTiming | hipEventDisableSystemFence));
        for (int i = 0; i < 3; i++) {
            HIP_TRY(hipStreamCreateWithFlags(&prog->bq_side[i], hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&prog->bq_join[i], hipEventDisableTiming | hipEventDisableSystemFence));
        }
    }
    const int lanes = (int)std::min<size_t>(ng, 4);          /* streams in use: st + lanes - 1 sides */
    HIP_TRY(hipEventRecord(prog->bq_fork, st));
    for (int k = 1; k < lanes; k++) HIP_TRY(hipStreamWaitEvent(prog->bq_side[k - 1], prog->bq_fork, 0));
    for (size_t gi = 0; gi < ng; gi++) {
        auto &g = *todo[gi];
        const int k = (int)(gi % (size_t)lanes);
        if (launch_group(g, k ? prog->bq_side[k - 1] : st, nullptr)) return -1;
    }
    for (int k = 1; k < lanes; k++) {
        HIP_TRY(hipEventRecord(prog->bq_join[k - 1], prog->bq_side[k - 1]));
        HIP_TRY(hipStreamWaitEvent(st, prog->bq_join[k - 1], 0));
    }
    if (last_stop) HIP_TRY(hipEventRecord(last_stop, st));
    return 0;
}

template <int FMT>
int launch_all(avdsp_hip_prog *prog, Plan &pl, BlockIO io, int fir_impl, int biquad_impl, hipStream_t stream)
{
    const bool under = prog->overlap && pl.overlap_ok && biquad_impl && fir_impl;
    pl.seq++;                                             /* this launch's number in the plan's ready words */
    if (under) {
        if (overlap_ready(prog)) return -1;
        const bool own_fir = prog->overlap >= 2 || prog->cu_split > 0;      /* the FIRs on a stream of the library's own */
        if (probe_side_by_side(prog, prog->overlap >= 2 ? prog->s_fir[prog->blk & 1] : prog->cu_split > 0 ? prog->s_fir[0] : stream)) return -1;     /* (remembered per stream) */
        /* fir_tile finds its cascades' blocks through the ready words; the other FIR kernels wait for the cascades' event */
        const bool can_words = (fir_impl == 1 || fir_impl == 4) && pl.d_ready && prog->d_ready_timeouts && prog->side_by_side;
        /* Mode 2 takes the wait packet off the FIRs' stream (a FIR follows the previous one like any kernel of a queue): 4096 chains
         * 0.5025 -> 0.4980 ms per step.  Where the cascades are the bound that packet's ~9 us are bubbles they live on: 2048 chains
         * 0.2594 -> 0.2686, 512 chains 0.0864 -> 0.0933 (one box) -- so by default only where a launch is more than one round of waves */
        /* (not with "overlap" 2: two FIRs in flight, the later one's waves would sit on their SIMDs looking at words of a cascade that
         * both of them starve -- 0.496 -> 0.547 ms) */
        const int rw = prog->ready_words >= 0 ? prog->ready_words : (prog->overlap == 1 && (long long)pl.n_fir * pl.max_taps >= 12000000ll ? 2 : 0);
        const bool behind = rw == 2 && can_words;                            /* a kernel behind the cascade publishes */
        const bool words = (rw == 1 && fir_impl == 1 && can_words) || behind;      /* the FIR looks at the words: no event wait on its stream */
        prog->ready_mode_now = behind ? 2 : words ? 1 : 0;
        const bool wt = words && !behind;                                  /* the cascade's own waves publish: write-through ring stores */
        const int slot = (int)(prog->blk % avdsp_hip_prog::kAhead);
        if (prog->ev_fir_set[slot]) {                           /* FIR k-3: the ring positions this block's cascade appends are free once it has read its window */
            /* The HOST waits for it ("ring_wait" 1, the default), not the cascades' stream.  A wait packet costs a queue ~9 us even when
             * its signal is long down, and the cascades' stream of a small shard has none to spare: a cascade beside an MFMA stream
             * lasts about a step, so with the packet in front of every cascade that stream's period -- not the FIRs' -- was the step
             * (512 chains: cascade 88.7 + packet ~9 = the 98 us measured; tools/step_gaps.py), and a run settled in that state or in
             * the one where the cascades are a block ahead (86 us) by chance.  The host is the natural place: it only has to stay
             * within three blocks of the device, and in steady state that IS this wait.  Bounded (1 ms), then the packet after all:
             * a caller whose stream waits on something it will enqueue later must not hang here. */
            bool done = false;
            if (prog->ring_wait_host) {
                const auto t0 = std::chrono::steady_clock::now();
                for (;;) {
                    const hipError_t q = hipEventQuery(prog->ev_fir_now[slot]);
                    if (q == hipSuccess) { done = true; break; }
                    (void)hipGetLastError();
                    if (q != hipErrorNotReady) break;
                    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(1)) break;
                    std::this_thread::yield();
                }
            }
            if (!done) HIP_TRY(hipStreamWaitEvent(prog->s_bq, prog->ev_fir_now[slot], 0));
        }
        if (prog->input_ready) HIP_TRY(hipStreamWaitEvent(prog->s_bq, prog->input_ready, 0));           /* (queued host blocks: the copy of this block) */
        if (launch_cascades<FMT>(prog, pl, io, biquad_impl, prog->s_bq, !words ? prog->ev_bq[slot] : nullptr, wt)) return -1;
        if (behind && prog->ready_test > 0) prog->ready_test--;       /* (tests: this launch's words are never set) */
        else if (behind) {
            hipLaunchKernelGGL(ready_set, dim3((unsigned)((pl.n_fir + kBlock - 1) / kBlock)), dim3(kBlock), 0, prog->s_bq, pl.d_ready, pl.d_fir_ids, pl.n_fir, pl.seq);
            HIP_TRY(hipGetLastError());
        }
        if (own_fir) {
            /* ("cu_split": ONE stream of the library's own, on the CUs the cascades' stream does not have)
             * the FIRs of consecutive blocks on two streams of the library's own, in turn: FIR k+1 needs nothing of FIR k, and on one
             * stream it would start a queue hand-over (~10 us) after FIR k's last wave; here its first workgroups fill the chip as FIR
             * k's last ones leave.  The caller's stream only waits for each FIR's end.  (The mode's contract then covers the output
             * too: the block a call writes must not be one an earlier call's FIR may still be writing.) */
            hipStream_t fs = prog->s_fir[prog->overlap >= 2 ? (prog->blk & 1) : 0];
            if (!words) HIP_TRY(hipStreamWaitEvent(fs, prog->ev_bq[slot], 0));
            if (launch_fir<FMT>(prog, pl, pl.d_fir_ids, pl.n_fir, io, fir_impl, fs, words)) return -1;
            HIP_TRY(hipEventRecord(prog->ev_fir[slot], fs));
            prog->ev_fir_now[slot] = prog->ev_fir[slot];
            HIP_TRY(hipStreamWaitEvent(stream, prog->ev_fir[slot], 0));
        } else {
            /* The FIR follows the previous block's FIR on the caller's stream with nothing in between: an event wait here -- a barrier
             * packet on another queue's signal -- holds the next dispatch back by ~9 us even when the signal is long down
             * (tools/step_gaps.py: 10.7 us between two FIRs of the 4096-chain program, 1.5-2 us between two kernels of one queue). */
            if (!words) HIP_TRY(hipStreamWaitEvent(stream, prog->ev_bq[slot], 0));
            /* (round 5, tools/shard_blocks_ab.sh: mode 2's empty microseconds only pay where the cascades are the clock, and that takes
             * a block long enough for them to be -- shards of 512 / 1024 chains, mode 2 against mode 1: 86.9 / 90.8 and 146.5 / 148.6 us at
             * 1024 frames, but 150.4 / 138.0 at 768, 91.2 / 78.1 and 57.6 / 51.6 at 512, 58.1 / 47.9 and 56.6 / 45.4 at 256) */
            const int mode = fir_impl != 1 && fir_impl != 4 ? 0 : prog->fir_launch_mode >= 0 ? prog->fir_launch_mode
                           : ((long long)pl.n_fir * pl.max_taps <= 6000000ll && io.nframes > 768) ? 2 : 1;
            /* (a launch whose kernel timer is sampled carries the timer's events instead; its event is then recorded behind it) */
            const bool rides = mode == 1 && !((prog->profile >> AVDSP_KERNEL_FIR & 1u) && prog->profile_seen[AVDSP_KERNEL_FIR & 7] % (unsigned)prog->profile_stride == 0);
            prog->fir_mode_now = mode;
            prog->last_ride_stop = nullptr;
            const int rc_fir = launch_fir<FMT>(prog, pl, pl.d_fir_ids, pl.n_fir, io, fir_impl, stream, words, rides ? prog->ev_fir[slot] : nullptr);
            prog->fir_mode_now = 0;
            if (rc_fir) return -1;
            /* (round 5: a launch that carries its kernel timer's stop event needs no second event behind it -- that one says "this FIR
             * has ended" just as well, and a recorded event is a marker packet the next dispatch queues behind: with every launch of a
             * 0.5-ms FIR timed the step was 0.5110 ms against 0.5045 with every fourth) */
            prog->ev_fir_now[slot] = prog->ev_fir[slot];
            static const bool timer_doubles = !getenv("AVDSP_TIMER_DOUBLES") || atoi(getenv("AVDSP_TIMER_DOUBLES")) != 0;
            if (!rides && mode == 1 && prog->last_ride_stop && timer_doubles) prog->ev_fir_now[slot] = prog->last_ride_stop;
            else if (!rides) HIP_TRY(hipEventRecord(prog->ev_fir[slot], stream));
        }
        prog->ev_fir_set[slot] = true;
        prog->blk++;
    } else {
        if (launch_cascades<FMT>(prog, pl, io, biquad_impl, stream, nullptr, false)) return -1;
        if (pl.n_fir && launch_fir<FMT>(prog, pl, pl.d_fir_ids, pl.n_fir, io, fir_impl, stream)) return -1;
    }
    if (pl.n_pass) {
        ProfileScope scope(prog, stream, AVDSP_KERNEL_PASS); scope.begin();
        PassArgs a{pl.d_chains, pl.d_pass_ids, pl.n_pass, io};
        const long long total = (long long)pl.n_pass * io.nframes;
        const int grid = (int)std::min<long long>((total + kBlock - 1) / kBlock, 2048);
        hipLaunchKernelGGL(passthrough<FMT>, dim3(grid), dim3(kBlock), 0, stream, a);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

}  // namespace

extern "C" {

const char *avdsp_hip_last_error(void) { return g_err; }

int avdsp_hip_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { set_err("hipGetDeviceCount: %s", hipGetErrorString(e)); return 0; }
    return n;
}

int avdsp_hip_set_device(int ordinal)
{
    HIP_TRY(hipSetDevice(ordinal));
    return 0;
}

avdsp_hip_prog *avdsp_hip_prog_create(int total_words)
{
    auto *p = new avdsp_hip_prog();
    p->total_words = total_words;
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) p->num_cus = cus;
    }
    /* + 2 words: the interpreter fetches the two words behind every head word, also behind the last one */
    hipError_t e = hipMalloc((void **)&p->d_buf, ((size_t)(total_words > 0 ? total_words : 1) + 2) * sizeof(int));
    if (e != hipSuccess) { set_err("hipMalloc(mirror, %d words): %s", total_words, hipGetErrorString(e)); delete p; return nullptr; }
    if (const char *m = getenv("AVDSP_FIR_LAUNCH_MODE")) p->fir_launch_mode = atoi(m);
    if (hipMalloc((void **)&p->d_ready_timeouts, 16) != hipSuccess || hipMemset(p->d_ready_timeouts, 0, 16) != hipSuccess) p->d_ready_timeouts = nullptr;   /* (without it the FIR waits for events) */
    if (p->d_ready_timeouts) {
        /* the word a timed-out wave sets, in mapped pinned host memory; without it no ready words at all (a time-out nobody hears of) */
        unsigned *dflag = nullptr;
        if (hipHostMalloc((void **)&p->h_ready_flag, 64, hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer((void **)&dflag, p->h_ready_flag, 0) == hipSuccess &&
            hipMemcpy(p->d_ready_timeouts + 2, &dflag, sizeof dflag, hipMemcpyHostToDevice) == hipSuccess) {
            *(volatile unsigned *)p->h_ready_flag = 0;
        } else {
            if (p->h_ready_flag) (void)hipHostFree(p->h_ready_flag);
            p->h_ready_flag = nullptr;
            (void)hipFree(p->d_ready_timeouts); p->d_ready_timeouts = nullptr;
            (void)hipGetLastError();
        }
    }
    return p;
}

void avdsp_hip_prog_destroy(avdsp_hip_prog *p)
{
    if (!p) return;
    (void)hipDeviceSynchronize();
    for (auto &pl : p->plans) free_plan(pl);
    for (auto &sp : p->spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    for (auto e : p->free_events) (void)hipEventDestroy(e);
    (void)hipFree(p->d_table); (void)hipHostFree(p->h_table);
    for (auto e : p->table_done) if (e) (void)hipEventDestroy(e);
    for (auto st : p->side) (void)hipStreamDestroy(st);
    for (auto e : p->join) (void)hipEventDestroy(e);
    if (p->fork) (void)hipEventDestroy(p->fork);
    (void)hipFree(p->d_tag_prev);
    for (auto &pn : p->pinned) if (pn.ours) (void)hipHostUnregister(const_cast<void *>(pn.ptr));
    for (auto st : {p->s_h2d, p->s_run, p->s_d2h}) if (st) (void)hipStreamDestroy(st);
    for (auto e : p->ev_host) (void)hipEventDestroy(e);
    for (auto st : {p->q_h2d, p->q_run, p->q_d2h}) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (auto &sl : p->hq) {
        (void)hipFree(sl.d_in); (void)hipFree(sl.d_out);
        for (auto e : {sl.h2d, sl.run, sl.d2h}) if (e) (void)hipEventDestroy(e);
    }
    if (p->s_bq) (void)hipStreamDestroy(p->s_bq);
    for (auto st : p->retired) (void)hipStreamDestroy(st);
    for (auto st : p->bq_side) if (st) (void)hipStreamDestroy(st);
    for (auto e : p->bq_join) if (e) (void)hipEventDestroy(e);
    if (p->bq_fork) (void)hipEventDestroy(p->bq_fork);
    for (auto fs : p->s_fir) if (fs) (void)hipStreamDestroy(fs);
    if (p->ev_unpack) (void)hipEventDestroy(p->ev_unpack);
    for (int i = 0; i < avdsp_hip_prog::kAhead; i++) { if (p->ev_bq[i]) (void)hipEventDestroy(p->ev_bq[i]); if (p->ev_fir[i]) (void)hipEventDestroy(p->ev_fir[i]); }
    (void)hipFree(p->d_buf); (void)hipFree(p->d_in); (void)hipFree(p->d_out); (void)hipFree(p->d_tpdf); (void)hipFree(p->d_frame);
    (void)hipHostFree(p->h_small); (void)hipHostFree(p->h_ready_flag);
    for (auto e : p->launch_ev) if (e) (void)hipEventDestroy(e);
    (void)hipFree(p->d_tpdf_seq); (void)hipFree(p->d_ready_timeouts); for (auto &x : p->alias) (void)hipFree(x.buf);
    (void)hipFree(p->d_inst_buf); (void)hipFree(p->d_inst_tpdf); (void)hipFree(p->d_inst_frame); (void)hipFree(p->d_inst_seq);
    delete p;
}

int avdsp_hip_prog_add_plan(avdsp_hip_prog *prog, const avdsp_plan_desc *d)
{
    if (d->format < 2 || d->format > 6) return set_err("format %d has no device kernels", d->format);
    Plan pl;
    pl.format = d->format; pl.nchains = d->nchains; pl.store_mask = d->store_mask;
    pl.lane_mode = d->format == 3 || d->format == 5;
    std::vector<avdsp_chain> chains(d->chains, d->chains + d->nchains);
    std::vector<int> coef(d->sec_coef_word, d->sec_coef_word + d->nsections);
    std::vector<int> state(d->sec_state_word, d->sec_state_word + d->nsections);
    /* every word index the kernels will touch must lie inside the mirror (chain instances: inside the copies, all of them) */
    const long long buf_words = d->instances > 1 ? (long long)AVDSP_INSTANCE_STRIDE(prog->total_words) * d->instances : (long long)prog->total_words;
    if (d->instances > 1 && d->instances != prog->chain_inst) return set_err("a plan of %d instances, the device holds %d copies of the program", d->instances, prog->chain_inst);
    pl.instances = d->instances > 1 ? d->instances : 1;
    for (int i = 0; i < d->nsections; i++)
        if (coef[i] < 0 || coef[i] + 5 > buf_words || state[i] < 0 || state[i] + 6 > buf_words || (state[i] & 1))
            return set_err("section %d addresses words outside the loaded buffer", i);
    std::vector<std::pair<int, std::vector<int>>> byN;   /* (section count, chain ids) in first-seen order */
    std::vector<int> fir, pass;
    pl.io_in_min = pl.io_out_min = 0x7FFFFFFF; pl.io_in_max = pl.io_out_max = -1;
    for (int i = 0; i < d->nchains; i++) {
        const avdsp_chain &c = chains[i];
        if (c.sec_base < 0 || c.nsec < 0 || c.sec_base + c.nsec > d->nsections) return set_err("chain %d: bad section range", i);
        if (c.n_out < 1 || c.n_out > AVDSP_MAX_STORES || c.in_io < 0) return set_err("chain %d: bad IO", i);
        pl.io_in_min = std::min(pl.io_in_min, c.in_io); pl.io_in_max = std::max(pl.io_in_max, c.in_io);
        for (int k = 0; k < c.n_out; k++) {
            if (c.out_io[k] < 0) return set_err("chain %d: bad IO", i);
            pl.io_out_min = std::min(pl.io_out_min, c.out_io[k]); pl.io_out_max = std::max(pl.io_out_max, c.out_io[k]);
        }
        if (c.fir_taps) {
            if (d->format == 2) return set_err("chain %d: FIR has no int64 definition", i);
            if (c.fir_coef_word < 0 || c.fir_coef_word + c.fir_taps > buf_words ||
                c.fir_state_word < 0 || c.fir_state_word + c.fir_taps > buf_words)
                return set_err("chain %d: FIR addresses words outside the loaded buffer", i);
            fir.push_back(i);
            pl.max_taps = std::max(pl.max_taps, c.fir_taps);
        }
        if (c.nsec) {
            auto it = std::find_if(byN.begin(), byN.end(), [&](const auto &e) { return e.first == c.nsec; });
            if (it == byN.end()) { byN.push_back({c.nsec, {}}); it = byN.end() - 1; }
            it->second.push_back(i);
        } else if (!c.fir_taps) pass.push_back(i);
    }
    if (upload_vec(&pl.d_sec_coef, coef) || upload_vec(&pl.d_sec_state, state)) { free_plan(pl); return -1; }
    if (pl.lane_mode && upload_vec(&pl.d_chains, chains)) { free_plan(pl); return -1; }
    std::vector<avdsp_chain> dev_chains = chains;        /* what the kernels see: the host's records + the pieces of long cascades (below) */
    if (pl.lane_mode) {                                  /* no launch groups, no rings: chain_lane walks the chain list itself */
        pl.n_lane_fir = (int)fir.size();             /* (n_fir stays 0: that one counts chains with a ring) */
        std::vector<int> rows;
        for (int i = 0; i < d->nchains; i++) if (chains[i].nsec >= 1 && chains[i].nsec <= 16) rows.push_back(i);
        pl.n_lane_rows = (int)rows.size();
        for (int i = 0; i < d->nchains; i++) pl.n_lane_feed += chains[i].nsec == 0 && chains[i].fir_taps != 0;
        if (upload_vec(&pl.d_lane_rows, rows)) { free_plan(pl); return -1; }
        prog->plans.push_back(pl);
        return (int)prog->plans.size() - 1;
    }
    std::vector<RowRec> all_rows; std::vector<LaneRec> all_lanes;
    for (auto &e : byN) {                                /* > 64 sections (P = 128): pieces of up to 64, below */
        /* lanes per chain: the next power of two -- but a 16-lane row per chain while the chip has SIMDs to spare
         * (<= 1024 waves): its step is shorter (one input batch per 16 steps, no mid-row section-0 lanes: cfg5's
         * 8-section cascades 85 -> 62 us) and idle lanes cost nothing there */
        int P = e.first > 64 ? 128 : pow2ceil(e.first);
        if (P < 16 && (long long)e.second.size() * 16 <= 65536) P = 16;
        bool all_fir = true;
        for (int id : e.second) all_fir = all_fir && chains[id].fir_taps != 0;
        Plan::Group g{P, e.first, (int)e.second.size(), nullptr, all_fir, nullptr, nullptr};
        if (upload_vec(&g.d_ids, e.second)) { free_plan(pl); return -1; }
        /* ... and in format 6 every cascade of more than 16 sections runs as pieces of up to 16, each a biquad_row launch: the word between two
         * sections is a float there and a plain LOAD / STORE pair moves it unchanged, so biquad_row takes the pieces as they are; a 16-lane row
         * per chain wastes no lanes on lengths like 17 or 33 (biquad_pipe: 32 / 64 lanes per chain) and its step is the shorter one --
         * 4096 chains x 17 / 33 / 48 / 65 / 200 sections: see DESIGN.md 4.1.  (Formats 2 and 4: pieces only beyond 64 sections, through biquad_pipe.) */
        const int piece_max = 16;
        if (e.first > piece_max) {
            /* A cascade of more than 64 sections does not fit a wave's lanes.  biquad_simple (a lane per chain, state in memory) took 81 ms
             * for 4096 chains x 65 sections against 136 us for 64: the chain is CUT instead, into pieces of equal length (+- 1) that run as
             * launches of biquad_pipe one after the other.  The pieces but the last are chain records of their own behind the host's
             * (no FIR, no SAT0DB, one raw store into the scratch column); the last piece is the chain's own record with its input moved
             * to the scratch column (its ring, stores and ready word are the chain's). */
            const int np = (e.first + piece_max - 1) / piece_max, base = e.first / np, extra = e.first % np, n = (int)e.second.size();
            int at = 0;
            for (int k = 0; k < np; k++) {
                const int len = base + (k < extra ? 1 : 0);
                std::vector<int> ids(n);
                for (int j = 0; j < n; j++) {
                    const avdsp_chain &c = chains[e.second[j]];
                    avdsp_chain pc = c;
                    pc.sec_base = c.sec_base + at; pc.nsec = len;
                    if (k > 0) { pc.in_io = j; pc.load_mode = kLoadRaw; }
                    if (k + 1 < np) {
                        pc.fir_taps = 0; pc.sat = kStoreRaw; pc.n_out = 1; pc.out_io[0] = j;
                        ids[j] = (int)dev_chains.size(); dev_chains.push_back(pc);
                    } else { ids[j] = e.second[j]; dev_chains[e.second[j]] = pc; }
                }
                Plan::Group pg{len <= 16 ? 16 : pow2ceil(len), len, n, nullptr, k + 1 == np && all_fir, nullptr, nullptr};
                pg.raw_out = k + 1 < np;
                auto drop = [&]() { (void)hipFree(g.d_ids); (void)hipFree(pg.d_ids); (void)hipFree(pg.d_rows); (void)hipFree(pg.d_lanes);
                                    for (auto &x : g.pieces) { (void)hipFree(x.d_ids); (void)hipFree(x.d_rows); (void)hipFree(x.d_lanes); } free_plan(pl); };
                if (upload_vec(&pg.d_ids, ids)) { drop(); return -1; }
                if (pg.P == 16) {                        /* biquad_row's records of the piece (format 6: kLoadRaw reads as a plain load there, and is one) */
                    std::vector<RowRec> rows(n);
                    std::vector<LaneRec> lanes((size_t)n * 16, LaneRec{-1, -1});
                    for (int j = 0; j < n; j++) {
                        const avdsp_chain &c = dev_chains[ids[j]];
                        /* bit 8: SAT0DB in front of the store -- in the int64 kernel "the stored word is acc >> 28", which is also what a piece
                         * hands on (bit 10 then takes the dither mask off); in the double kernels a piece's word is the float as it is: no bit 8 */
                        const bool raw = c.sat == kStoreRaw;
                        rows[j] = RowRec{ids[j], c.in_io, c.out_io[0], (c.load_mode & 0xFF) | ((c.sat == 1 || (raw && d->format == 2)) ? 1 << 8 : 0) | (c.fir_taps ? 1 << 9 : 0) |
                                         (raw ? 1 << 10 : 0) | (c.n_out << 16), c.gain_bits, {c.nsec, 0, 0}};
                        for (int q = 0; q < c.nsec; q++) lanes[(size_t)j * 16 + (16 - c.nsec) + q] = LaneRec{coef[c.sec_base + q], state[c.sec_base + q]};
                    }
                    if (upload_vec(&pg.d_rows, rows) || upload_vec(&pg.d_lanes, lanes)) { drop(); return -1; }
                }
                g.pieces.push_back(pg);
                at += len;
            }
            for (int k = 0; k < 2; k++)
                if (hipMalloc((void **)&g.d_scratch[k], (size_t)kFirChunk * n * sizeof(unsigned)) != hipSuccess) {
                    (void)hipFree(g.d_ids); (void)hipFree(g.d_scratch[0]); for (auto &x : g.pieces) { (void)hipFree(x.d_ids); (void)hipFree(x.d_rows); (void)hipFree(x.d_lanes); } free_plan(pl);
                    return set_err("hipMalloc(scratch of %d long cascades)", n);
                }
        }
        if (P == 16 && g.pieces.empty()) {               /* what biquad_row loads instead of walking group -> chain -> section tables */
            std::vector<RowRec> rows(e.second.size());
            std::vector<LaneRec> lanes(e.second.size() * 16, LaneRec{-1, -1});
            for (size_t i = 0; i < e.second.size(); i++) {
                const avdsp_chain &c = chains[e.second[i]];
                rows[i] = RowRec{e.second[i], c.in_io, c.out_io[0], (c.load_mode & 0xFF) | (c.sat ? 1 << 8 : 0) | (c.fir_taps ? 1 << 9 : 0) | (c.n_out << 16),
                                 c.gain_bits, {c.nsec, 0, 0}};
                for (int k = 0; k < c.nsec; k++) lanes[i * 16 + (16 - c.nsec) + k] = LaneRec{coef[c.sec_base + k], state[c.sec_base + k]};
            }
            if (upload_vec(&g.d_rows, rows) || upload_vec(&g.d_lanes, lanes)) { (void)hipFree(g.d_ids); (void)hipFree(g.d_rows); free_plan(pl); return -1; }
            /* ... and the same rows in the table of all lengths: this run, filled up to whole waves */
            all_rows.insert(all_rows.end(), rows.begin(), rows.end());
            all_lanes.insert(all_lanes.end(), lanes.begin(), lanes.end());
            /* (an empty row is a copy of a real one with no chain behind it: every lane of a wave FETCHES, section or not -- its input
             * column must be one the block has; nothing of it is stored) */
            RowRec empty = rows[0]; empty.cid = -1;
            while (all_rows.size() % 4) { all_rows.push_back(empty); all_lanes.insert(all_lanes.end(), 16, LaneRec{-1, -1}); }
            pl.n_row_groups++;
            pl.rows_all_fir = pl.rows_all_fir && all_fir;
        }
        pl.bq.push_back(g);
    }
    if (upload_vec(&pl.d_chains, dev_chains)) { free_plan(pl); return -1; }
    if (pl.n_row_groups >= 2) {
        RowRec empty = all_rows.back(); empty.cid = -1;
        while (all_rows.size() % 16) { all_rows.push_back(empty); all_lanes.insert(all_lanes.end(), 16, LaneRec{-1, -1}); }
        pl.n_rows_all = (int)all_rows.size();
        if (upload_vec(&pl.d_rows_all, all_rows) || upload_vec(&pl.d_lanes_all, all_lanes)) { free_plan(pl); return -1; }
    }
    pl.n_fir = (int)fir.size(); pl.n_pass = (int)pass.size();
    if (upload_vec(&pl.d_fir_ids, fir) || upload_vec(&pl.d_pass_ids, pass)) { free_plan(pl); return -1; }
    if (pl.n_fir) {
        pl.fir_gpc = fir_groups_per_chunk(pl.max_taps);
        {   /* nothing in the launch path may touch function attributes (stream capture) */
            int hs_cap, row;
            const int lds = (int)fir_lds_bytes(pl.fir_gpc, &hs_cap, &row);
            const void *variants[2] = { d->format == 4 ? (const void *)fir_mfma<4, 1> : (const void *)fir_mfma<6, 1>,
                                        d->format == 4 ? (const void *)fir_mfma<4, 2> : (const void *)fir_mfma<6, 2> };
            for (const void *fn : variants) {
                hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
                if (e != hipSuccess) { free_plan(pl); return set_err("hipFuncSetAttribute(LDS %d): %s", lds, hipGetErrorString(e)); }
            }
        }
        /* + one more launch of frames: under "overlap" the cascade appends block k+1 while the FIR still reads block k's window */
        pl.ring_R = pow2ceil(pl.max_taps + avdsp_hip_prog::kAhead * kFirChunk + 16 * pl.fir_gpc + 16 * (kNG + 4) + 64);
        static_assert(kFirChunk == kFirPad, "one FIR launch covers exactly the frames the window image is laid out for");
        hipError_t e = hipMalloc((void **)&pl.d_ring, (size_t)d->nchains * 2 * pl.ring_R * sizeof(float));      /* (every sample twice: Ring) */
        if (e != hipSuccess) { free_plan(pl); return set_err("hipMalloc(FIR rings, %d x %d): %s", d->nchains, pl.ring_R, hipGetErrorString(e)); }
        pl.wpos = 0;
        for (int i = 0; i < d->nchains; i++) pl.n_fir_only += chains[i].fir_taps && !chains[i].nsec;
        {
            const void *fns[3] = { d->format == 4 ? (const void *)fir_stream<4, 1> : (const void *)fir_stream<6, 1>,
                                   d->format == 4 ? (const void *)fir_stream<4, 2> : (const void *)fir_stream<6, 2>,
                                   d->format == 4 ? (const void *)fir_stream<4, 4> : (const void *)fir_stream<6, 4> };
            const int flds[3] = { 4 * StreamGeom<1>::LDS_DOUBLES * 8 + 64, 4 * StreamGeom<2>::LDS_DOUBLES * 8 + 64, 4 * StreamGeom<4>::LDS_DOUBLES * 8 + 64 };
            for (int v = 0; v < 3; v++) {
                hipError_t e2 = hipFuncSetAttribute(fns[v], hipFuncAttributeMaxDynamicSharedMemorySize, flds[v]);
                if (e2 != hipSuccess) { free_plan(pl); return set_err("hipFuncSetAttribute(fir_stream LDS %d): %s", flds[v], hipGetErrorString(e2)); }
            }
            const void *ffn[4] = { d->format == 4 ? (const void *)fir_flow<4, 1> : (const void *)fir_flow<6, 1>,
                                   d->format == 4 ? (const void *)fir_flow<4, 2> : (const void *)fir_flow<6, 2>,
                                   d->format == 4 ? (const void *)fir_flow<4, 4> : (const void *)fir_flow<6, 4>,
                                   d->format == 4 ? (const void *)fir_flow<4, 1, true> : (const void *)fir_flow<6, 1, true> };
            const int fl[4] = { 4 * FlowGeom<1>::LDS_DOUBLES * 8 + 64, 4 * FlowGeom<2>::LDS_DOUBLES * 8 + 64, 4 * FlowGeom<4>::LDS_DOUBLES * 8 + 64,
                                4 * FlowGeom<1, true>::LDS_DOUBLES * 8 + 64 };
            for (int v = 0; v < 4; v++) {
                hipError_t e2 = hipFuncSetAttribute(ffn[v], hipFuncAttributeMaxDynamicSharedMemorySize, fl[v]);
                if (e2 != hipSuccess) { free_plan(pl); return set_err("hipFuncSetAttribute(fir_flow LDS %d): %s", fl[v], hipGetErrorString(e2)); }
            }
        }
        {   /* fir_tile: LDS opt-in per variant, and the taps as doubles */
            {
                const bool f4 = d->format == 4;
                const void *more[6] = { f4 ? (const void *)fir_tile<4, 1, false, true> : (const void *)fir_tile<6, 1, false, true>,
                                        f4 ? (const void *)fir_tile<4, 1, false, true, false> : (const void *)fir_tile<6, 1, false, true, false>,
                                        f4 ? (const void *)fir_tile<4, 1, false, false, false> : (const void *)fir_tile<6, 1, false, false, false>,
                                        f4 ? (const void *)fir_tile<4, 2, false, false, false> : (const void *)fir_tile<6, 2, false, false, false>,
                                        f4 ? (const void *)fir_tile<4, 4, false, false, false> : (const void *)fir_tile<6, 4, false, false, false>,
                                        f4 ? (const void *)fir_tile<4, 1, true, false, false> : (const void *)fir_tile<6, 1, true, false, false> };
                const int mlds[6] = { 4 * TileGeom<1>::LDS_DOUBLES * 8 + 64, 4 * TileGeom<1>::LDS_DOUBLES * 8 + 64, 4 * TileGeom<1>::LDS_DOUBLES * 8 + 64,
                                      4 * TileGeom<2>::LDS_DOUBLES * 8 + 64, 4 * TileGeom<4>::LDS_DOUBLES * 8 + 64, 4 * TileGeom<1, true>::LDS_DOUBLES * 8 + 64 };
                for (int v = 0; v < 6; v++) {
                    hipError_t e3 = hipFuncSetAttribute(more[v], hipFuncAttributeMaxDynamicSharedMemorySize, mlds[v]);
                    if (e3 != hipSuccess) { free_plan(pl); return set_err("hipFuncSetAttribute(fir_tile LDS): %s", hipGetErrorString(e3)); }
                }
            }
            const void *tiles[4] = { d->format == 4 ? (const void *)fir_tile<4, 1> : (const void *)fir_tile<6, 1>,
                                     d->format == 4 ? (const void *)fir_tile<4, 2> : (const void *)fir_tile<6, 2>,
                                     d->format == 4 ? (const void *)fir_tile<4, 4> : (const void *)fir_tile<6, 4>,
                                     d->format == 4 ? (const void *)fir_tile<4, 1, true> : (const void *)fir_tile<6, 1, true> };
            const int tlds[4] = { 4 * TileGeom<1>::LDS_DOUBLES * 8 + 64, 4 * TileGeom<2>::LDS_DOUBLES * 8 + 64, 4 * TileGeom<4>::LDS_DOUBLES * 8 + 64,
                                  4 * TileGeom<1, true>::LDS_DOUBLES * 8 + 64 };
            for (int v = 0; v < 4; v++) {
                hipError_t e2 = hipFuncSetAttribute(tiles[v], hipFuncAttributeMaxDynamicSharedMemorySize, tlds[v]);
                if (e2 != hipSuccess) { free_plan(pl); return set_err("hipFuncSetAttribute(fir_tile LDS %d): %s", tlds[v], hipGetErrorString(e2)); }
            }
            pl.pitch64 = taps64_pitch(pl.max_taps);
            hipError_t e2 = hipMalloc((void **)&pl.d_taps64, (size_t)d->nchains * pl.pitch64 * sizeof(double));
            if (e2 != hipSuccess) { free_plan(pl); return set_err("hipMalloc(f64 taps, %d x %d): %s", d->nchains, pl.pitch64, hipGetErrorString(e2)); }
            Taps64Args ta{prog->d_buf, pl.d_chains, pl.d_fir_ids, pl.d_taps64, pl.pitch64};
            hipLaunchKernelGGL(taps_to_f64, dim3(pl.n_fir), dim3(kBlock), 0, nullptr, ta);
            if (hipGetLastError() != hipSuccess) { free_plan(pl); return set_err("taps_to_f64 failed to launch"); }
        }
        RingConvArgs ca{prog->d_buf, pl.d_chains, pl.d_fir_ids, pl.n_fir, plan_ring(pl)};
        hipLaunchKernelGGL(state_to_ring, dim3(pl.n_fir), dim3(kBlock), 0, nullptr, ca);   /* history the caller's buffer holds */
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) { free_plan(pl); return set_err("state_to_ring failed"); }
    }
    {
        long long nout = 0;
        for (int i = 0; i < d->nchains; i++) nout += chains[i].n_out;           /* check_independent (host): no IO is stored twice */
        pl.stores_whole_window = pl.io_out_max >= pl.io_out_min && nout == (long long)pl.io_out_max - pl.io_out_min + 1;
    }
    pl.overlap_ok = pl.n_fir > 0 && !pl.bq.empty();
    for (int i = 0; i < d->nchains && pl.overlap_ok; i++)
        if (chains[i].nsec && !chains[i].fir_taps) pl.overlap_ok = false;
    if (pl.overlap_ok) {                                  /* ready words, all at launch number 0 */
        if (hipMalloc((void **)&pl.d_ready, (size_t)d->nchains * sizeof(unsigned)) != hipSuccess ||
            hipMemset(pl.d_ready, 0, (size_t)d->nchains * sizeof(unsigned)) != hipSuccess) { free_plan(pl); return set_err("hipMalloc(ready words)"); }
    }
    prog->plans.push_back(pl);
    return (int)prog->plans.size() - 1;
}

/* LDS budget of the generic path: frame (when small) + staged mirror, one workgroup per CU at most */
static const int kGenericFrameLds = 4096;            /* words */
static const int kGenericBatchLds = 8192;            /* words: input + output rows of one batch of frames */
static const size_t kGenericLdsMax = 144 * 1024;     /* bytes */

int avdsp_hip_prog_add_generic(avdsp_hip_prog *prog, const avdsp_generic_desc *d)
{
    if (d->format < 2 || d->format > 6) return set_err("format %d is not one of 2..6", d->format);
    if (d->core_word < 0 || d->core_word >= d->prog_words || d->prog_words > prog->total_words || d->io_span < 1 ||
        (d->end_word && (d->end_word <= d->core_word || d->end_word > d->prog_words)))
        return set_err("generic plan: core word %d / program %d words / IO span %d do not fit the mirror (%d words)",
                       d->core_word, d->prog_words, d->io_span, prog->total_words);
    if (!prog->d_tpdf) return set_err("generic plan before avdsp_hip_tpdf_reset");
    Plan pl;
    pl.generic = true; pl.format = d->format; pl.io_span = d->io_span;
    pl.io_in_min = d->io_in_min; pl.io_in_max = d->io_in_max; pl.io_out_min = d->io_out_min; pl.io_out_max = d->io_out_max;
    GenericArgs &a = pl.ga;
    a.buf = prog->d_buf; a.tpdf = prog->d_tpdf;
    a.core_word = d->core_word; a.prog_words = d->prog_words; a.end_word = d->end_word;
    if (d->skip_from && (d->skip_from <= d->core_word || d->skip_to <= d->skip_from || d->skip_to >= d->prog_words))
        return set_err("generic plan: skipped stretch [%d, %d) outside the piece", d->skip_from, d->skip_to);
    a.skip_from = d->skip_from; a.skip_to = d->skip_to;
    a.freq_index = d->freq_index; a.num_freq = d->num_freq;
    a.biquad_skip = d->biquad_freq_skip; a.biquad_offset = d->biquad_freq_offset;
    a.delay_factor = d->delay_line_factor;
    /* one samples[] frame per program, shared by its cores */
    const int want = std::max(d->io_span, kGenericFrameLds);
    if (prog->frame_words < want) {
        /* a later core with a wider IO span: the frame grows, keeps its content, and the earlier cores' plans follow it */
        unsigned *grown = nullptr;
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMalloc((void **)&grown, (size_t)want * 4));
        HIP_TRY(hipMemset(grown, 0, (size_t)want * 4));
        if (prog->d_frame) HIP_TRY(hipMemcpy(grown, prog->d_frame, (size_t)prog->frame_words * 4, hipMemcpyDeviceToDevice));
        (void)hipFree(prog->d_frame);
        prog->d_frame = grown; prog->frame_words = want;
        for (auto &o : prog->plans) if (o.generic) o.ga.scratch = grown;
    }
    a.scratch = prog->d_frame;
    a.scratch_len = prog->frame_words;
    /* staged: LDS = [frame][mirror][batch rows], everything the interpreter touches per opcode is LDS;
     * otherwise (program or IO span too large) LDS = [batch rows] and it works out of HBM */
    const size_t staged_bytes = ((size_t)prog->frame_words + prog->total_words + kGenericBatchLds) * 4;
    pl.ga_staged = staged_bytes <= kGenericLdsMax;
    if (pl.ga_staged) {
        a.frame_lds = prog->frame_words;
        a.stage_words = prog->total_words; a.keep_words = (int)(sizeof(dspHeader_t) / 4);
        a.batch_lds = a.frame_lds + a.stage_words;
        pl.ga_lds = staged_bytes;
    } else {
        a.frame_lds = 0; a.stage_words = 0; a.batch_lds = 0;
        pl.ga_lds = (size_t)kGenericBatchLds * 4;
    }
    /* what the core owns (written back after a launch); unknown = everything, and the core then runs alone */
    a.nown = -1; a.own = nullptr; a.tpdf_owner = d->tpdf_calc != 0;
    a.tpdf_role = d->tpdf_role; a.tpdf_seq = nullptr;
    pl.dither_only = d->dither_only != 0 && d->dither_result_word >= d->prog_words && d->dither_result_word + ((d->format == 3 || d->format == 5) ? 1 : 2) <= prog->total_words;
    pl.dither_arg = d->dither_arg; pl.dither_word = d->dither_result_word;
    a.nrd_slot = a.nwr_slot = -1;
    if (d->nown >= 0 && d->io_span <= 256) {
        std::vector<int> own(d->own, d->own + 2 * (size_t)d->nown);
        for (int i = 0; i < d->nown; i++)
            if (own[2 * i] < 0 || own[2 * i + 1] < own[2 * i] || own[2 * i + 1] > prog->total_words)
                return set_err("generic plan: owned range %d outside the mirror", i);
        if (upload_vec(&pl.d_own, own)) return -1;
        a.own = pl.d_own; a.nown = d->nown;
        for (int k = 0; k < 8; k++) a.written_io[k] = d->written_io[k];
        int nrd = 0, nwr = 0;
        for (int sl = 0; sl < 256; sl++) {
            if (d->early_io[sl >> 5] >> (sl & 31) & 1u) { if (nrd < 32) a.rd_slot[nrd] = (unsigned char)sl; nrd++; }
            if (d->written_io[sl >> 5] >> (sl & 31) & 1u) { if (nwr < 32) a.wr_slot[nwr] = (unsigned char)sl; nwr++; }
        }
        if (nrd <= 32 && nwr <= 32) { a.nrd_slot = nrd; a.nwr_slot = nwr; }
    }
    pl.wave_ok = d->wave_ok != 0 && d->nvm >= 0 && d->nvm <= 16;
    if (pl.wave_ok) {
        for (int k = 0; k < 8; k++) pl.carried_io[k] = d->carried_io[k];
        a.nvm = d->nvm;
        for (int k = 0; k < d->nvm; k++) a.vm_word[k] = d->vm_word[k];
        a.seq_words = std::max(d->seq_words, 256);      /* also parks the persistent frame (<= 256 slots) at start */
        const void *fn = d->format == 2 ? (const void *)interp_wave<2> : d->format == 3 ? (const void *)interp_wave<3>
                       : d->format == 4 ? (const void *)interp_wave<4> : d->format == 5 ? (const void *)interp_wave<5>
                                                                                        : (const void *)interp_wave<6>;
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGenericLdsMax);
        if (e != hipSuccess) { (void)hipFree(pl.d_own); return set_err("hipFuncSetAttribute(frame-parallel LDS): %s", hipGetErrorString(e)); }
        const void *gfn = d->format == 2 ? (const void *)interp_wave_grid<2> : d->format == 3 ? (const void *)interp_wave_grid<3>
                        : d->format == 4 ? (const void *)interp_wave_grid<4> : d->format == 5 ? (const void *)interp_wave_grid<5>
                                                                                          : (const void *)interp_wave_grid<6>;
        e = hipFuncSetAttribute(gfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGenericLdsMax);
        if (e != hipSuccess) { (void)hipFree(pl.d_own); return set_err("hipFuncSetAttribute(frame-parallel LDS): %s", hipGetErrorString(e)); }
        const void *ifn = d->format == 2 ? (const void *)interp_wave_instances<2> : d->format == 3 ? (const void *)interp_wave_instances<3>
                        : d->format == 4 ? (const void *)interp_wave_instances<4> : d->format == 5 ? (const void *)interp_wave_instances<5>
                                                                                          : (const void *)interp_wave_instances<6>;
        e = hipFuncSetAttribute(ifn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGenericLdsMax);
        if (e != hipSuccess) { (void)hipFree(pl.d_own); return set_err("hipFuncSetAttribute(frame-parallel LDS): %s", hipGetErrorString(e)); }
    }
    if (pl.ga_staged) {   /* per plan creation, like the FIR: nothing in the launch path may touch function attributes */
        const void *fn = d->format == 2 ? (const void *)interp_core<2, true> : d->format == 3 ? (const void *)interp_core<3, true>
                       : d->format == 4 ? (const void *)interp_core<4, true> : d->format == 5 ? (const void *)interp_core<5, true>
                                                                                              : (const void *)interp_core<6, true>;
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGenericLdsMax);
        if (e != hipSuccess) { (void)hipFree(pl.d_own); return set_err("hipFuncSetAttribute(generic LDS): %s", hipGetErrorString(e)); }
    }
    prog->plans.push_back(pl);
    return (int)prog->plans.size() - 1;
}

static int plan_add_strands(avdsp_hip_prog *prog, Plan &pl, const avdsp_strand_desc *d);

/* A strand plan is an extra on a generic plan that is complete without it: when it cannot be attached (a shape the kernel's tables do
 * not hold, an offset that fails the checks below, a device that refuses the LDS request), nothing of it stays behind -- the plan then
 * runs through the interpreter as if it had never been asked -- and the caller learns why from the return value and the message. */
int avdsp_hip_plan_add_strands(avdsp_hip_prog *prog, int plan, const avdsp_strand_desc *d)
{
    if (plan < 0 || plan >= (int)prog->plans.size() || !prog->plans[plan].generic) return set_err("strand plan: %d is not a generic plan", plan);
    Plan &pl = prog->plans[plan];
    const int rc = plan_add_strands(prog, pl, d);
    if (rc) {
        (void)hipFree(pl.d_sops); (void)hipFree(pl.d_sargs);
        pl.d_sops = nullptr; pl.d_sargs = nullptr;
        pl.s_loaded.clear(); pl.s_stored.clear();
        pl.s_nops = pl.s_nargs = pl.s_nstrands = pl.s_nres = 0; pl.s_usey = false;
    }
    return rc;
}

static int plan_add_strands(avdsp_hip_prog *prog, Plan &pl, const avdsp_strand_desc *d)
{
    const int aw = (pl.format == 3 || pl.format == 5) ? 1 : 2;      /* words of an accumulator in memory */
    if (d->nops < 1 || d->nstrands < 1 || d->nargs < 1) return set_err("strand plan: empty");
    if (d->nops > kStrandMaxOps) return set_err("strand plan: %d operations per strand (the kernel's list holds %d)", d->nops, kStrandMaxOps);
    std::vector<avdsp_strand_op> ops(d->ops, d->ops + d->nops);
    std::vector<int> args(d->args, d->args + (size_t)d->nstrands * d->nargs);
    /* every word index and IO number the kernel will follow lies inside the mirror / the frame (the host derived them from offsets
     * scan_generic has checked; checked once more here, like the chain plans) */
    for (auto &o : ops) {
        const int cols[3] = {o.a0, o.a1, o.a2};
        for (int c : cols) if (c < 0 || c >= d->nargs) return set_err("strand plan: argument column %d outside the row", c);
        for (int r = 0; r < d->nstrands; r++) {
            const int *row = args.data() + (size_t)r * d->nargs;
            auto word_ok = [&](int w, int n) { return w >= 0 && (long long)w + n <= prog->total_words; };
            bool ok = true;
            switch (o.op) {
            case AVDSP_SOP_LOAD: ok = row[o.a0] >= 0 && row[o.a0] < prog->frame_words; pl.s_loaded.push_back(row[o.a0]); break;
            case AVDSP_SOP_LOAD_GAIN: ok = row[o.a0] >= 0 && row[o.a0] < prog->frame_words && word_ok(row[o.a1], 1); pl.s_loaded.push_back(row[o.a0]); break;
            case AVDSP_SOP_STORE: ok = row[o.a0] >= 0 && row[o.a0] < prog->frame_words; pl.s_stored.push_back(row[o.a0]); break;
            case AVDSP_SOP_GAIN: case AVDSP_SOP_SAT0DB_GAIN: case AVDSP_SOP_SAT0DB_TPDF_GAIN: ok = word_ok(row[o.a0], 1); break;
            case AVDSP_SOP_LOAD_MEM: case AVDSP_SOP_STORE_MEM: ok = word_ok(row[o.a0], aw); break;
            case AVDSP_SOP_DELAY: case AVDSP_SOP_DELAY_DP:
                {   /* the line: the first word is its size in samples when a parameter word gives the delay, else microseconds */
                    const long long nline = row[o.a2] ? (long long)row[o.a0] : (long long)(((unsigned long long)(unsigned)row[o.a0] * pl.ga.delay_factor) >> 32);
                    ok = row[o.a0] >= 0 && word_ok(row[o.a1], 1) && (long long)row[o.a1] + 1 + nline * (o.op == AVDSP_SOP_DELAY_DP ? aw : 1) <= prog->total_words &&
                         (row[o.a2] == 0 || word_ok(row[o.a2], 1));
                }
                break;
            case AVDSP_SOP_BIQUADS:
                ok = o.imm >= 1 && word_ok(row[o.a0], 6 * o.imm) && word_ok(row[o.a1], pl.ga.biquad_offset + (o.imm - 1) * pl.ga.biquad_skip + 5);
                break;
            default: break;
            }
            if (!ok) return set_err("strand plan: strand %d addresses words or IOs outside the loaded buffer", r);
        }
    }
    if (upload_vec(&pl.d_sops, ops) || upload_vec(&pl.d_sargs, args)) return -1;
    {
        int nres = 0;
        for (auto &o : ops) { if (o.rcol != nres) return set_err("strand plan: resolved columns are not laid out in order"); nres += avdsp_strand_rcols(o.op, o.imm, aw); }
        if (nres != d->nres || nres > 224) return set_err("strand plan: %d resolved columns (the kernel's table holds 224)", nres);
        pl.s_usey = false;                               /* does any operation read Y?  (else the kernel does not keep it) */
        for (auto &o : ops)
            pl.s_usey = pl.s_usey || o.op == AVDSP_SOP_COPYYX || o.op == AVDSP_SOP_SWAPXY || o.op == AVDSP_SOP_ADDXY || o.op == AVDSP_SOP_ADDYX ||
                        o.op == AVDSP_SOP_SUBXY || o.op == AVDSP_SOP_SUBYX;
        const void *fns[10] = {(const void *)strand_lanes<2, false>, (const void *)strand_lanes<3, false>, (const void *)strand_lanes<4, false>,
                               (const void *)strand_lanes<5, false>, (const void *)strand_lanes<6, false>,
                               (const void *)strand_lanes<2, true>, (const void *)strand_lanes<3, true>, (const void *)strand_lanes<4, true>,
                               (const void *)strand_lanes<5, true>, (const void *)strand_lanes<6, true>};
        /* the attribute is the function's ceiling: raised to what the largest plan seen so far needs (a launch asks for its own
         * plan's), never to the tables' limit -- a part with less LDS than that limit still runs the plans that fit */
        static int lds_ceilings[64][10];                 /* [device][function]: the attribute is set per device */
        int dev_now = 0;
        HIP_TRY(hipGetDevice(&dev_now));
        int *lds_ceiling = lds_ceilings[dev_now & 63];
        const int fi = pl.format - 2 + (pl.s_usey ? 5 : 0);
        const int need = std::max(nres, 1) * 256 + 8192 + 64 + d->nops * (int)sizeof(avdsp_strand_op);
        if (need > lds_ceiling[fi]) {
            const hipError_t e = hipFuncSetAttribute(fns[fi], hipFuncAttributeMaxDynamicSharedMemorySize, need);
            if (e != hipSuccess) return set_err("hipFuncSetAttribute(strand_lanes LDS, %d bytes): %s", need, hipGetErrorString(e));
            lds_ceiling[fi] = need;
        }
    }
    pl.s_nops = d->nops; pl.s_nargs = d->nargs; pl.s_nstrands = d->nstrands; pl.s_nres = d->nres;
    return 0;
}

int avdsp_hip_plan_strands(const avdsp_hip_prog *prog, int plan)
{
    return prog && plan >= 0 && plan < (int)prog->plans.size() ? prog->plans[plan].s_nstrands : 0;
}

/* does the strand kernel take this call's windows?  (Else the stretch runs through the interpreter, which handles every case.)
 * The windows must not share IO numbers; a strand must not store into the input window (the next frame's input would replace the
 * value, the interpreter keeps that order) nor load, outside the input window, an IO of the output window (the caller's row shows
 * through there). */
static bool strands_take(const Plan &pl, const BlockIO &io)
{
    if (!pl.s_nstrands) return false;
    const bool overlap = io.in_stride > 0 && io.out_stride > 0 && io.in_base < io.out_base + io.out_stride && io.out_base < io.in_base + io.in_stride;
    if (overlap) return false;
    for (int s : pl.s_stored) if (s >= io.in_base && s < io.in_base + io.in_stride) return false;
    for (int l : pl.s_loaded)
        if (!(l >= io.in_base && l < io.in_base + io.in_stride) && l >= io.out_base && l < io.out_base + io.out_stride) return false;
    return true;
}

int avdsp_hip_tpdf_reset(avdsp_hip_prog *prog, int seed, int default_dither)
{
    if (!prog->d_tpdf) HIP_TRY(hipMalloc((void **)&prog->d_tpdf, sizeof(TpdfGlobals)));
    hipLaunchKernelGGL(tpdf_init_kernel, dim3(1), dim3(1), 0, nullptr, prog->d_tpdf, seed, default_dither);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return 0;
}

/* The frame-parallel kernel runs the block when the core allows it (plan) and this call's windows do:
 * every slot the core reads before storing must come from the caller, and the per-lane frames must fit */
static bool wave_plan_fits(const avdsp_hip_prog *prog, const Plan &pl, const BlockIO &io, GenericArgs &a)
{
    if (!pl.wave_ok || io.nframes < 2) return false;     /* a single frame (dspRuntime_N) has nothing to put side by side */
    /* with slot lists the rows move slot by slot, so the columns only need to reach the core's own highest slot: a
     * piece that touches little IO (the dither prefix of a wide core) stays frame-parallel whatever the windows' width */
    const int span = a.rows_whole ? std::max(pl.io_span, std::max(io.in_base + io.in_stride, io.out_base + io.out_stride)) : pl.io_span;
    if (span > 256) return false;
    for (int s = 0; s < 256; s++)
        if (pl.carried_io[s >> 5] >> (s & 31) & 1u) {
            const bool in_win = s >= io.in_base && s < io.in_base + io.in_stride;
            const bool out_win = s >= io.out_base && s < io.out_base + io.out_stride;
            if (!in_win && !out_win) return false;
        }
    const size_t words = (size_t)span * interp::kLanePitch + prog->total_words + 2 + 128 + (size_t)a.nvm * 128 + a.seq_words;
    if (words * 4 > kGenericLdsMax) return false;
    a.wave_span = span;
    a.frame_lds = span * interp::kLanePitch;
    a.stage_words = prog->total_words; a.keep_words = (int)(sizeof(dspHeader_t) / 4);
    a.batch_lds = a.frame_lds + a.stage_words + 2;
    return true;
}

/* where the two windows share IO numbers the host loop hands the input through to the output rows for slots the
 * core does not store: that (and unknown ownership) needs the whole windows moved, not the core's slots */
static bool windows_overlap(const BlockIO &io)
{
    return io.in_stride > 0 && io.out_stride > 0 && io.in_base < io.out_base + io.out_stride && io.out_base < io.in_base + io.in_stride;
}
static bool needs_whole(const Plan &pl) { return pl.ga.nown < 0 || pl.ga.nrd_slot < 0 || pl.ga.nwr_slot < 0; }
static bool rows_whole(const Plan &pl, const BlockIO &io) { return windows_overlap(io) || needs_whole(pl); }

/* ... unless the handing-through is done apart (round 4): one small kernel copies the shared IO numbers' columns from the input rows
 * to the output rows; every launch behind it then moves only its core's slots (a slot in both windows is READ from the input row
 * and STORED to the output row, so the copy and the launches do not meet), and the pieces of a level can run side by side again.
 * This is the reference's one samples[] frame: an IO inside both windows shows the input unless somebody stores it.  (The reference's
 * own dacdiy1.bin has its outputs on both sides of its inputs: no pair of windows that take all of them can be kept apart.) */
static __global__ void show_through(const unsigned *in, int in_stride, int in_base, unsigned *out, int out_stride, int out_base,
                             int nframes, int lo, int hi, size_t in_inst_words, size_t out_inst_words)
{
    const int w = hi - lo;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)nframes * w) return;
    const int f = (int)(idx / w), sl = lo + (int)(idx % w);
    out[(size_t)blockIdx.y * out_inst_words + (size_t)f * out_stride + (sl - out_base)] =
        in[(size_t)blockIdx.y * in_inst_words + (size_t)f * in_stride + (sl - in_base)];
}
static int launch_show_through(const BlockIO &io, hipStream_t stream, int ninst = 1, size_t in_inst_words = 0, size_t out_inst_words = 0)
{
    const int lo = std::max(io.in_base, io.out_base), hi = std::min(io.in_base + io.in_stride, io.out_base + io.out_stride);
    if (hi <= lo || io.nframes <= 0) return 0;
    const long long total = (long long)io.nframes * (hi - lo);
    hipLaunchKernelGGL(show_through, dim3((unsigned)((total + kBlock - 1) / kBlock), (unsigned)ninst), dim3(kBlock), 0, stream,
                       io.in, io.in_stride, io.in_base, io.out, io.out_stride, io.out_base, io.nframes, lo, hi, in_inst_words, out_inst_words);
    HIP_TRY(hipGetLastError());
    return 0;
}
/* a dspRuntimeBlockAll call does it once, in front of its first level (on the caller's stream: every launch of the call is ordered behind it) */
struct ShownScope {
    avdsp_hip_prog *p;
    explicit ShownScope(avdsp_hip_prog *prog) : p(prog) { p->call_shown = true; }
    ~ShownScope() { p->call_shown = false; }
};

/* pieces of a cut core exchange the per-frame dither values through one buffer per program */
static int tpdf_seq_for(avdsp_hip_prog *prog, GenericArgs &a, int nframes)
{
    if (!a.tpdf_role) return 0;
    if (prog->tpdf_seq_frames < nframes) {
        HIP_TRY(hipDeviceSynchronize());
        (void)hipFree(prog->d_tpdf_seq); prog->d_tpdf_seq = nullptr; prog->tpdf_seq_frames = 0;
        const int cap = std::max(nframes, 4096);
        HIP_TRY(hipMalloc((void **)&prog->d_tpdf_seq, (size_t)cap * 2 * sizeof(int)));
        HIP_TRY(hipMemset(prog->d_tpdf_seq, 0, (size_t)cap * 2 * sizeof(int)));
        prog->tpdf_seq_frames = cap;
    }
    a.tpdf_seq = prog->d_tpdf_seq;
    return 0;
}

static int launch_generic(avdsp_hip_prog *prog, Plan &pl, BlockIO io, hipStream_t stream)
{
    GenericArgs a = pl.ga;
    a.io = io;
    a.rows_whole = rows_whole(pl, io);
    if (tpdf_seq_for(prog, a, io.nframes)) return -1;
    if (windows_overlap(io) && !needs_whole(pl) && !strands_take(pl, io)) {
        /* behind the copy of the shared columns either interpreter kernel delivers its core's slots only.  (Both: inside a
         * dspRuntimeBlockAll call the pieces of a level run side by side, and a frame-by-frame piece that wrote whole rows back would
         * overwrite what its neighbours store meanwhile -- found by the overlapping-window sweep, tests/dev/gpu_wave_sweep.py.) */
        if (!prog->call_shown && launch_show_through(io, stream)) return -1;
        a.rows_whole = 0;
    }
    if (strands_take(pl, io)) {
        ProfileScope scope(prog, stream, AVDSP_KERNEL_STRANDS); scope.begin();
        StrandArgs sa{};
        sa.buf = a.buf; sa.tpdf = a.tpdf; sa.tpdf_seq = a.tpdf_seq; sa.tpdf_role = a.tpdf_role;
        sa.ops = pl.d_sops; sa.nops = pl.s_nops; sa.args = pl.d_sargs; sa.nargs = pl.s_nargs; sa.nstrands = pl.s_nstrands; sa.nres = pl.s_nres;
        const size_t slds = (size_t)std::max(pl.s_nres, 1) * 256 + 8192 + 64 + (size_t)pl.s_nops * sizeof(avdsp_strand_op);   /* per-lane table, delay exchange area, dither values, operation list */
        sa.prog_words = a.prog_words; sa.biquad_skip = a.biquad_skip; sa.biquad_offset = a.biquad_offset; sa.delay_factor = a.delay_factor;
        sa.scratch = a.scratch; sa.io = io;
#ifdef AVDSP_BQ_STAMPS
        {
            static unsigned long long *d_st = nullptr;
            if (!d_st) { HIP_TRY(hipMalloc((void **)&d_st, 32 * 8)); HIP_TRY(hipMemset(d_st, 0, 32 * 8)); }
            sa.stamps = d_st; g_bq_stamps = d_st; g_bq_stamp_waves = 1;
        }
#endif
        const dim3 sgrid((pl.s_nstrands + 63) / 64), sblock(64);
#define AVDSP_LAUNCH_STRANDS(F) \
        if (pl.s_usey) hipLaunchKernelGGL((strand_lanes<F, true>), sgrid, sblock, slds, stream, sa); \
        else           hipLaunchKernelGGL((strand_lanes<F, false>), sgrid, sblock, slds, stream, sa)
        switch (pl.format) {
        case 2:  AVDSP_LAUNCH_STRANDS(2); break;
        case 3:  AVDSP_LAUNCH_STRANDS(3); break;
        case 4:  AVDSP_LAUNCH_STRANDS(4); break;
        case 5:  AVDSP_LAUNCH_STRANDS(5); break;
        default: AVDSP_LAUNCH_STRANDS(6); break;
        }
#undef AVDSP_LAUNCH_STRANDS
        HIP_TRY(hipGetLastError());
        return 0;
    }
    const dim3 grid(1), block(64);
    if (pl.dither_only && pl.wave_ok && io.nframes > 1) {      /* (wave_ok: frame-parallel forms not switched off) */
        ProfileScope scope(prog, stream, AVDSP_KERNEL_GENERIC_WAVE); scope.begin();
        switch (pl.format) {
        case 2:  hipLaunchKernelGGL((tpdf_walk<2>), grid, block, 0, stream, a, pl.dither_arg, pl.dither_word, io.nframes); break;
        case 3:  hipLaunchKernelGGL((tpdf_walk<3>), grid, block, 0, stream, a, pl.dither_arg, pl.dither_word, io.nframes); break;
        case 4:  hipLaunchKernelGGL((tpdf_walk<4>), grid, block, 0, stream, a, pl.dither_arg, pl.dither_word, io.nframes); break;
        case 5:  hipLaunchKernelGGL((tpdf_walk<5>), grid, block, 0, stream, a, pl.dither_arg, pl.dither_word, io.nframes); break;
        default: hipLaunchKernelGGL((tpdf_walk<6>), grid, block, 0, stream, a, pl.dither_arg, pl.dither_word, io.nframes); break;
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (wave_plan_fits(prog, pl, io, a)) {
        ProfileScope scope(prog, stream, AVDSP_KERNEL_GENERIC_WAVE); scope.begin();
        const size_t lds = ((size_t)a.batch_lds + 128 + (size_t)a.nvm * 128 + a.seq_words) * 4;
        switch (pl.format) {
        case 2:  hipLaunchKernelGGL((interp_wave<2>), grid, block, lds, stream, a); break;
        case 3:  hipLaunchKernelGGL((interp_wave<3>), grid, block, lds, stream, a); break;
        case 4:  hipLaunchKernelGGL((interp_wave<4>), grid, block, lds, stream, a); break;
        case 5:  hipLaunchKernelGGL((interp_wave<5>), grid, block, lds, stream, a); break;
        default: hipLaunchKernelGGL((interp_wave<6>), grid, block, lds, stream, a); break;
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }
    ProfileScope scope(prog, stream, AVDSP_KERNEL_GENERIC); scope.begin();
    if (prog->done_offer) { a.done_flag = prog->done_offer; a.done_seq = prog->small_seq; prog->done_taken = true; }      /* (the single-frame call's wait) */
#define AVDSP_LAUNCH_INTERP(F) \
    if (pl.ga_staged) hipLaunchKernelGGL((interp_core<F, true>), grid, block, pl.ga_lds, stream, a); \
    else              hipLaunchKernelGGL((interp_core<F, false>), grid, block, pl.ga_lds, stream, a)
    switch (pl.format) {
    case 2:  AVDSP_LAUNCH_INTERP(2); break;
    case 3:  AVDSP_LAUNCH_INTERP(3); break;
    case 4:  AVDSP_LAUNCH_INTERP(4); break;
    case 5:  AVDSP_LAUNCH_INTERP(5); break;
    default: AVDSP_LAUNCH_INTERP(6); break;
    }
#undef AVDSP_LAUNCH_INTERP
    HIP_TRY(hipGetLastError());
    return 0;
}

static int check_range(avdsp_hip_prog *p, int first, int n)
{
    if (first < 0 || n < 0 || first + n > p->total_words) return set_err("word range [%d,%d) outside the mirror (%d words)", first, first + n, p->total_words);
    return 0;
}

/* The FIR delay lines live in rings on the device; the mirror's state words are brought up to date
 * before they are read back, and the rings are rebuilt after the mirror's state words were written. */
static int rings_to_mirror(avdsp_hip_prog *p)
{
    for (auto &pl : p->plans)
        if (pl.n_fir) {
            RingConvArgs ca{p->d_buf, pl.d_chains, pl.d_fir_ids, pl.n_fir, plan_ring(pl)};
            hipLaunchKernelGGL(ring_to_state, dim3(pl.n_fir), dim3(kBlock), 0, nullptr, ca);
            HIP_TRY(hipGetLastError());
        }
    return 0;
}

static int mirror_to_rings(avdsp_hip_prog *p)
{
    for (auto &pl : p->plans)
        if (pl.n_fir) {
            RingConvArgs ca{p->d_buf, pl.d_chains, pl.d_fir_ids, pl.n_fir, plan_ring(pl)};
            hipLaunchKernelGGL(state_to_ring, dim3(pl.n_fir), dim3(kBlock), 0, nullptr, ca);
            HIP_TRY(hipGetLastError());
        }
    return 0;
}

int avdsp_hip_prog_clear_plans(avdsp_hip_prog *p)
{
    HIP_TRY(hipDeviceSynchronize());
    if (rings_to_mirror(p)) return -1;
    HIP_TRY(hipDeviceSynchronize());
    for (auto &pl : p->plans) free_plan(pl);
    p->plans.clear();
    return 0;
}

int avdsp_hip_upload_words(avdsp_hip_prog *p, const int32_t *host_buf, int first, int n)
{
    if (check_range(p, first, n)) return -1;
    HIP_TRY(hipDeviceSynchronize());
    if (n && copy_from_caller(p->d_buf + first, host_buf + first, (size_t)n * 4)) return -1;
    return mirror_to_rings(p);
}

int avdsp_hip_download_words(avdsp_hip_prog *p, int32_t *host_buf, int first, int n)
{
    if (check_range(p, first, n)) return -1;
    HIP_TRY(hipDeviceSynchronize());
    READY_CHECK(p);                                      /* the state the caller asks for would not be the reference's */
    if (rings_to_mirror(p)) return -1;
    HIP_TRY(hipDeviceSynchronize());
    if (n && copy_to_caller(host_buf + first, p->d_buf + first, (size_t)n * 4)) return -1;
    return 0;
}

int avdsp_hip_zero_words(avdsp_hip_prog *p, int first, int n)
{
    if (check_range(p, first, n)) return -1;
    HIP_TRY(hipDeviceSynchronize());
    if (n) HIP_TRY(hipMemset(p->d_buf + first, 0, (size_t)n * 4));
    return mirror_to_rings(p);
}

int avdsp_hip_run_block(avdsp_hip_prog *prog, int plan, const void *d_in, int in_stride, int in_io_base,
                        void *d_out, int out_stride, int out_io_base, int nframes,
                        int fir_impl, int biquad_impl, void *stream)
{
    if (plan < 0 || plan >= (int)prog->plans.size()) return set_err("bad plan id %d", plan);
    Plan &pl = prog->plans[plan];
    if (nframes <= 0) return 0;
    READY_CHECK(prog);                                   /* a time-out of an earlier block: this call fails, and every one after it */
    /* the chain kernels index the sample blocks with the chains' IO numbers: check the windows once here.
     * (The interpreter keeps a whole samples[] frame: slots outside the caller's windows are the frame's
     * own, persistent like the host's array, e.g. values one strand leaves for the next frame.) */
    /* (a plan of chain instances carries the instances' block offsets in its chains' IO numbers -- the host has checked the windows
     * against the program's own IOs) */
    if (!pl.generic && pl.instances <= 1 && pl.io_in_max >= pl.io_in_min && (pl.io_in_min < in_io_base || pl.io_in_max >= in_io_base + in_stride))
        return set_err("input window IO [%d,%d) does not cover the IOs the core loads [%d,%d]", in_io_base, in_io_base + in_stride, pl.io_in_min, pl.io_in_max);
    if (!pl.generic && pl.instances <= 1 && pl.io_out_max >= pl.io_out_min && (pl.io_out_min < out_io_base || pl.io_out_max >= out_io_base + out_stride))
        return set_err("output window IO [%d,%d) does not cover the IOs the core stores [%d,%d]", out_io_base, out_io_base + out_stride, pl.io_out_min, pl.io_out_max);
    if (pl.generic) {
        /* the scratch frame is indexed by IO number: both windows must lie inside it */
        if (in_stride < 0 || out_stride < 0 || in_io_base < 0 || out_io_base < 0 ||
            (in_stride && in_io_base + in_stride > pl.ga.scratch_len) || (out_stride && out_io_base + out_stride > pl.ga.scratch_len))
            return set_err("sample windows [%d,%d) / [%d,%d) reach past the core's IO span %d", in_io_base, in_io_base + in_stride,
                           out_io_base, out_io_base + out_stride, pl.ga.scratch_len);
        BlockIO io;
        io.in = (const unsigned *)d_in;  io.in_stride = in_stride;   io.in_base = in_io_base;
        io.out = (unsigned *)d_out;      io.out_stride = out_stride; io.out_base = out_io_base;
        io.nframes = nframes; io.store_mask = -1;
        if (in_stride + out_stride > kGenericBatchLds)
            return set_err("sample windows of %d + %d words per frame exceed the interpreter's batch buffer (%d)", in_stride, out_stride, kGenericBatchLds);
        pl.ga.batch_frames = std::max(1, std::min(64, kGenericBatchLds / std::max(1, in_stride + out_stride)));
        return launch_generic(prog, pl, io, (hipStream_t)stream);
    }
    /* In-place calls: input and output windows in the same memory (the usual way to process a block where it lies -- the IO numbers
     * differ, the columns coincide).  The block kernels read a frame before they store it, but a cascade that met an Inf / NaN /
     * huge exponent runs its block AGAIN from the input (the replays of biquad_row, biquad_pipe, chain_rows), and would then filter
     * its own outputs: chains that store straight from the cascade get their input from a copy made first (a FIR stores after the
     * cascades of its launch are through; the int64 model has no replay). */
    {
        const char *i0 = (const char *)d_in, *i1 = i0 + (size_t)nframes * in_stride * 4;
        const char *o0 = (const char *)d_out, *o1 = o0 + (size_t)nframes * out_stride * 4;
        const bool direct = pl.format != 2 && (pl.lane_mode || (!pl.bq.empty() && !pl.overlap_ok));
        if (direct && nframes > 1 && i0 < o1 && o0 < i1) {
            const size_t words = (size_t)nframes * in_stride;
            /* one copy buffer per caller's stream: a buffer is only ever written and read in that stream's order (two in-place calls on
             * different streams used to share one -- a race); growing one waits for its own stream, not for the device */
            avdsp_hip_prog::Alias *al = nullptr;
            for (auto &x : prog->alias) if (x.stream == (hipStream_t)stream) al = &x;
            if (!al) { prog->alias.push_back({(hipStream_t)stream, nullptr, 0}); al = &prog->alias.back(); }
            if (words > al->cap) {
                HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
                (void)hipFree(al->buf); al->buf = nullptr; al->cap = 0;
                HIP_TRY(hipMalloc((void **)&al->buf, words * 4)); al->cap = words;
            }
            HIP_TRY(hipMemcpyAsync(al->buf, d_in, words * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
            d_in = al->buf;
        }
    }
    if (pl.lane_mode) {
        ProfileScope scope(prog, (hipStream_t)stream, AVDSP_KERNEL_BIQUAD); scope.begin();
        LaneArgs a{};
        a.buf = prog->d_buf; a.chains = pl.d_chains; a.sec_coef = pl.d_sec_coef; a.sec_state = pl.d_sec_state; a.nchains = pl.nchains;
        a.io.in = (const unsigned *)d_in;  a.io.in_stride = in_stride;   a.io.in_base = in_io_base;
        a.io.out = (unsigned *)d_out;      a.io.out_stride = out_stride; a.io.out_base = out_io_base;
        a.io.nframes = nframes; a.io.store_mask = pl.store_mask;
        const dim3 grid((pl.nchains + 63) / 64), block(64);
        hipStream_t st = (hipStream_t)stream;
        /* blocks: the FIRs frame-parallel behind the cascades (fir_lane); single frames keep the tap loop on the delay line */
        const bool tiles = pl.n_lane_fir > 0 && nframes > 1;
        if (tiles) {
            const int pitch = (pl.max_taps - 1 + nframes + 63) / 64 * 64;
            if (pitch > pl.lseq_pitch) {
                HIP_TRY(hipDeviceSynchronize());
                (void)hipFree(pl.d_lseq); pl.d_lseq = nullptr; pl.lseq_pitch = 0;
                HIP_TRY(hipMalloc((void **)&pl.d_lseq, (size_t)pl.nchains * pitch * sizeof(unsigned)));
                pl.lseq_pitch = pitch;
            }
            a.seq = pl.d_lseq; a.pitch = pl.lseq_pitch; a.hist = pl.max_taps - 1;
            if (pl.max_taps > 1) hipLaunchKernelGGL(fir_lane_history, dim3(pl.nchains, (pl.max_taps - 1 + 255) / 256), dim3(256), 0, st, a);
        }
        /* blocks: cascades of up to 16 sections with a lane per section (chain_rows); the rest, and single frames, a lane per chain */
        const bool rows = pl.n_lane_rows > 0 && nframes > 1;
        a.rows_take = rows ? 1 : 0;
        a.lane_hw = prog->lane_hw;
        if (rows) {
            const dim3 rgrid((pl.n_lane_rows + 3) / 4);
            if (pl.format == 3) hipLaunchKernelGGL(chain_rows<3>, rgrid, block, 0, st, a, (const int *)pl.d_lane_rows, pl.n_lane_rows);
            else                hipLaunchKernelGGL(chain_rows<5>, rgrid, block, 0, st, a, (const int *)pl.d_lane_rows, pl.n_lane_rows);
        }
        if (tiles && pl.n_lane_feed > 0) {
            const dim3 fg(pl.nchains, (nframes + 255) / 256);
            if (pl.format == 3) hipLaunchKernelGGL(fir_lane_feed<3>, fg, dim3(256), 0, st, a);
            else                hipLaunchKernelGGL(fir_lane_feed<5>, fg, dim3(256), 0, st, a);
        }
        if ((rows ? pl.n_lane_rows : 0) + (tiles ? pl.n_lane_feed : 0) < pl.nchains) {      /* the rest: longer cascades, chains without filters, single frames */
            if (pl.format == 3) hipLaunchKernelGGL(chain_lane<3>, grid, block, 0, st, a);
            else                hipLaunchKernelGGL(chain_lane<5>, grid, block, 0, st, a);
        }
        if (tiles && prog->lane_hw) {
            /* the hardware's toward-zero product where it is the reference's (fir_lane_hw), two frames per lane */
            const dim3 fgrid(pl.nchains, (nframes + kFirHwFrames - 1) / kFirHwFrames), fblock(256);
            if (pl.format == 3) hipLaunchKernelGGL(fir_lane_hw<3>, fgrid, fblock, 0, st, a);
            else                hipLaunchKernelGGL(fir_lane_hw<5>, fgrid, fblock, 0, st, a);
            hipLaunchKernelGGL(fir_lane_state, dim3(pl.nchains, (pl.max_taps + 255) / 256), dim3(256), 0, st, a);
        } else if (tiles) {
            const dim3 fgrid(pl.nchains, (nframes + kFirLaneFrames - 1) / kFirLaneFrames), fblock(kFirLaneFrames);
            if (pl.format == 3) hipLaunchKernelGGL(fir_lane<3>, fgrid, fblock, 0, st, a);
            else                hipLaunchKernelGGL(fir_lane<5>, fgrid, fblock, 0, st, a);
            hipLaunchKernelGGL(fir_lane_state, dim3(pl.nchains, (pl.max_taps + 255) / 256), dim3(256), 0, st, a);
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }
    for (int f0 = 0; f0 < nframes; f0 += kFirChunk) {
        BlockIO io;
        io.in = (const unsigned *)d_in + (size_t)f0 * in_stride;  io.in_stride = in_stride;   io.in_base = in_io_base;
        io.out = (unsigned *)d_out + (size_t)f0 * out_stride;     io.out_stride = out_stride; io.out_base = out_io_base;
        io.nframes = std::min(kFirChunk, nframes - f0);
        io.store_mask = pl.store_mask;
        int rc;
        switch (pl.format) {
        case 2:  rc = launch_all<2>(prog, pl, io, fir_impl, biquad_impl, (hipStream_t)stream); break;
        case 4:  rc = launch_all<4>(prog, pl, io, fir_impl, biquad_impl, (hipStream_t)stream); break;
        default: rc = launch_all<6>(prog, pl, io, fir_impl, biquad_impl, (hipStream_t)stream); break;
        }
        if (rc) return rc;
        if (pl.n_fir) pl.wpos = (pl.wpos + io.nframes) & (pl.ring_R - 1);
    }
    return 0;
}

/* "host_pin": pin the caller's buffer where it lies (a host hands over the same buffers block after block): copies from and
 * to pinned memory are true DMA and run beside kernels.  Registrations are remembered, which is only sound while the caller
 * keeps those buffers allocated -- a registration outlives free(), and a new allocation at the same address would then
 * receive DMA into the OLD pages.  Hence opt-in; without it the synchronous calls copy through the driver's own staging and
 * the queued calls (avdsp_hip_submit_block_host) register a block's buffers for the time the block is in flight only --
 * avdsp_hip_wait_block_host releases them as it lets the block through, so a buffer may be freed once its block is back.   */
static void unpin_block(avdsp_hip_prog *prog, const void *ptr);
static void pin_in_place(avdsp_hip_prog *prog, const void *ptr, size_t bytes, bool hold = false)
{
    auto find = [&]() -> long { for (size_t i = 0; i < prog->pinned.size(); i++) if (prog->pinned[i].ptr == ptr) return (long)i; return -1; };
    long at = find();
    if (at >= 0 && prog->pinned[at].bytes >= bytes) { prog->pinned[at].refs += hold; return; }
    if (at >= 0) {
        /* the same address with more bytes than were registered: one entry per address -- the old registration goes (once no queued
         * block copies through it any more) and the buffer is registered again at its new size */
        if (prog->pinned[at].refs > 0) { (void)avdsp_hip_wait_block_host(prog, 0); at = find(); }      /* (the wait may have dropped the entry) */
        if (at >= 0) {
            if (prog->pinned[at].ours) (void)hipHostUnregister(const_cast<void *>(ptr));
            prog->pinned.erase(prog->pinned.begin() + at);
        }
    }
    if (prog->pinned.size() >= 16) {
        /* the table is full: registrations nobody holds go; those of queued blocks still copying (and the one a caller of this
         * function has just taken for its other buffer) stay */
        bool idle = false;
        for (auto &pn : prog->pinned) idle = idle || pn.refs == 0;
        if (!idle) (void)avdsp_hip_wait_block_host(prog, 0);
        for (size_t i = 0; i < prog->pinned.size();) {
            if (prog->pinned[i].refs > 0) { i++; continue; }
            if (prog->pinned[i].ours) (void)hipHostUnregister(const_cast<void *>(prog->pinned[i].ptr));
            prog->pinned.erase(prog->pinned.begin() + (long)i);
        }
    }
    const hipError_t e = hipHostRegister(const_cast<void *>(ptr), bytes, hipHostRegisterDefault);
    (void)hipGetLastError();
    prog->pinned.push_back({ptr, bytes, e == hipSuccess, hold ? 1 : 0});
}

/* a queued block has gone back: without "host_pin" its buffers' registrations end with it (the caller may free them now) */
static void unpin_block(avdsp_hip_prog *prog, const void *ptr)
{
    for (size_t i = 0; i < prog->pinned.size(); i++) {
        auto &pn = prog->pinned[i];
        if (pn.ptr != ptr) continue;
        if (pn.refs > 0) pn.refs--;
        if (pn.refs == 0 && !prog->host_pin) {
            if (pn.ours) (void)hipHostUnregister(const_cast<void *>(pn.ptr));
            prog->pinned.erase(prog->pinned.begin() + (long)i);
        }
        return;
    }
}

int avdsp_hip_run_block_host(avdsp_hip_prog *prog, int plan, const void *h_in, int in_stride, int in_io_base,
                             void *h_out, int out_stride, int out_io_base, int nframes,
                             int fir_impl, int biquad_impl)
{
    if (plan < 0 || plan >= (int)prog->plans.size()) return set_err("bad plan id %d", plan);
    Plan &pl = prog->plans[plan];
    if (in_stride == 0 && out_stride == 0) {
        /* single-frame dspRuntime_N(): both windows are the caller's samples[] array, IO 0 .. span */
        in_stride = out_stride = pl.generic ? pl.io_span : std::max(pl.io_in_max, pl.io_out_max) + 1;
        in_io_base = out_io_base = 0;
    }
    const size_t in_words = (size_t)nframes * in_stride, out_words = (size_t)nframes * out_stride;
    if (prog->in_cap < in_words) {
        (void)hipFree(prog->d_in); prog->d_in = nullptr; prog->in_cap = 0;
        HIP_TRY(hipMalloc((void **)&prog->d_in, in_words * 4)); prog->in_cap = in_words;
    }
    if (prog->out_cap < out_words) {
        (void)hipFree(prog->d_out); prog->d_out = nullptr; prog->out_cap = 0;
        HIP_TRY(hipMalloc((void **)&prog->d_out, out_words * 4)); prog->out_cap = out_words;
    }
    /* The chain kernels over a block of some size: the host loop of linux/avdsp_plugin.c:98-141 as a three-stage pipeline --
     * piece k+1 crosses PCIe while piece k is computed and piece k-1 goes back.  The output block is uploaded first only if
     * the core leaves slots of the window untouched (they must keep the caller's content).
     * Only over PINNED memory ("host_pin", or buffers the caller has pinned himself): asynchronous copies of pageable memory make
     * the runtime pin the caller's pages behind his back (copy_from_caller); pageable buffers take the synchronous way below. */
    const bool whole_window = !pl.generic && pl.stores_whole_window && out_io_base == pl.io_out_min && out_stride == pl.io_out_max - pl.io_out_min + 1;
    if (!pl.generic && nframes >= 256 && (prog->host_pin || (caller_memory_is_pinned(h_in) && caller_memory_is_pinned(h_out)))) {
        if (!prog->s_h2d) {
            HIP_TRY(hipStreamCreateWithFlags(&prog->s_h2d, hipStreamNonBlocking));
            HIP_TRY(hipStreamCreateWithFlags(&prog->s_run, hipStreamNonBlocking));
            HIP_TRY(hipStreamCreateWithFlags(&prog->s_d2h, hipStreamNonBlocking));
        }
        HIP_TRY(hipDeviceSynchronize());                    /* earlier work of any stream has finished: the pipeline starts clean */
        /* (held for the call: pinning the second buffer may have to make room in the table, and must not drop the first) */
        struct PinHold { avdsp_hip_prog *p; const void *a, *b; ~PinHold() { if (a) unpin_block(p, a); if (b) unpin_block(p, b); } } pin_hold{prog, nullptr, nullptr};
        if (prog->host_pin) {
            pin_in_place(prog, h_in, in_words * 4, true);   pin_hold.a = h_in;
            pin_in_place(prog, h_out, out_words * 4, true); pin_hold.b = h_out;
        }
        if (!whole_window) HIP_TRY(hipMemcpyAsync(prog->d_out, h_out, out_words * 4, hipMemcpyHostToDevice, prog->s_h2d));
        const int split = prog->host_split > 0 ? std::max(prog->host_split, 64) : nframes;
        const int npieces = (nframes + split - 1) / split;
        while ((int)prog->ev_host.size() < 2 * npieces) {
            hipEvent_t e; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            prog->ev_host.push_back(e);
        }
        for (int k = 0; k < npieces; k++) {
            const int f0 = k * split, nf = std::min(split, nframes - f0);
            const unsigned *hi = (const unsigned *)h_in + (size_t)f0 * in_stride;
            unsigned *di = prog->d_in + (size_t)f0 * in_stride, *dout = prog->d_out + (size_t)f0 * out_stride;
            HIP_TRY(hipMemcpyAsync(di, hi, (size_t)nf * in_stride * 4, hipMemcpyHostToDevice, prog->s_h2d));
            HIP_TRY(hipEventRecord(prog->ev_host[2 * k], prog->s_h2d));
            HIP_TRY(hipStreamWaitEvent(prog->s_run, prog->ev_host[2 * k], 0));
            prog->input_ready = prog->ev_host[2 * k];       /* "overlap": the cascade runs on a stream of its own and must wait for this piece's copy too */
            const int rc = avdsp_hip_run_block(prog, plan, di, in_stride, in_io_base, dout, out_stride, out_io_base, nf, fir_impl, biquad_impl, prog->s_run);
            prog->input_ready = nullptr;
            if (rc) return -1;
            HIP_TRY(hipEventRecord(prog->ev_host[2 * k + 1], prog->s_run));
            HIP_TRY(hipStreamWaitEvent(prog->s_d2h, prog->ev_host[2 * k + 1], 0));
            HIP_TRY(hipMemcpyAsync((unsigned *)h_out + (size_t)f0 * out_stride, dout, (size_t)nf * out_stride * 4, hipMemcpyDeviceToHost, prog->s_d2h));
        }
        HIP_TRY(hipStreamSynchronize(prog->s_d2h));
        HIP_TRY(hipStreamSynchronize(prog->s_run));
        HIP_TRY(hipStreamSynchronize(prog->s_h2d));      /* (nothing of this call stays behind on any stream) */
        READY_CHECK(prog);                               /* (synchronous: a time-out inside THIS block is this call's error) */
        return 0;
    }
    /* A frame or a few (dspRuntime_N: one): three synchronous copies would be most of the call.  The samples go through a small
     * pinned area the kernels read and write in place over PCIe -- a handful of words each way. */
    if (in_words + out_words <= avdsp_hip_prog::kSmallWords) {
        if (!prog->h_small) {
            HIP_TRY(hipHostMalloc((void **)&prog->h_small, (size_t)(avdsp_hip_prog::kSmallWords + 16) * 4, hipHostMallocMapped));
            HIP_TRY(hipHostGetDevicePointer((void **)&prog->d_small, prog->h_small, 0));
            prog->h_small[avdsp_hip_prog::kSmallWords] = 0; prog->small_seq = 0;
        }
        unsigned *hi = prog->h_small, *ho = prog->h_small + in_words;
        memcpy(hi, h_in, in_words * 4);
        memcpy(ho, h_out, out_words * 4);                   /* unstored slots keep their content */
        static const int flag_us = getenv("AVDSP_SMALL_FLAG") ? atoi(getenv("AVDSP_SMALL_FLAG")) : 2000;
        const bool use_flag = flag_us > 0 && !prog->overlap;
        if (use_flag) { ++prog->small_seq; prog->done_offer = prog->d_small + avdsp_hip_prog::kSmallWords; prog->done_taken = false; }
        const int rc_run = avdsp_hip_run_block(prog, plan, prog->d_small, in_stride, in_io_base, prog->d_small + in_words, out_stride, out_io_base,
                                               nframes, fir_impl, biquad_impl, nullptr);
        prog->done_offer = nullptr;
        if (rc_run) return -1;
        /* the wait: the "frame done" word (frame_done above) -- not under the overlap mode, whose kernels are on streams of their own */
        bool seen = false;
        if (use_flag) {
            const unsigned seq = prog->small_seq;
            if (!prog->done_taken) {                         /* (the frame's kernels are several, or of a kind that does not set the word itself) */
                hipLaunchKernelGGL(frame_done, dim3(1), dim3(1), 0, nullptr, prog->d_small + avdsp_hip_prog::kSmallWords, seq);
                HIP_TRY(hipGetLastError());
            }
            volatile unsigned *flag = prog->h_small + avdsp_hip_prog::kSmallWords;
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned spins = 0; !(seen = *flag == seq); spins++)
                if ((spins & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(flag_us)) break;
            std::atomic_thread_fence(std::memory_order_acquire);
        }
        if (!seen) HIP_TRY(hipDeviceSynchronize());        /* (a long frame, or no word: the runtime's own wait) */
        READY_CHECK(prog);
        memcpy(h_out, ho, out_words * 4);
        return 0;
    }
    if (copy_from_caller(prog->d_in, h_in, in_words * 4)) return -1;
    if (!whole_window && copy_from_caller(prog->d_out, h_out, out_words * 4)) return -1;   /* unstored slots keep their content */
    if (avdsp_hip_run_block(prog, plan, prog->d_in, in_stride, in_io_base, prog->d_out, out_stride, out_io_base,
                            nframes, fir_impl, biquad_impl, nullptr)) return -1;
    HIP_TRY(hipDeviceSynchronize());
    READY_CHECK(prog);                                   /* (synchronous: a time-out inside THIS block is this call's error) */
    if (copy_to_caller(h_out, prog->d_out, out_words * 4)) return -1;
    return 0;
}

/* The host-pointer block call as a queue: submit returns when the block's copies and kernels are enqueued, and up to
 * kHostQueue blocks are in flight -- block k+1 crosses PCIe while block k is computed and block k-1 goes back, which the
 * synchronous call (one block, three stages, one after the other) cannot do.  The caller's buffers are pinned in place
 * (hipHostRegister) on first use and must stay allocated and untouched until avdsp_hip_wait_block_host has let the block
 * through; blocks complete in submission order.  Cores that run through the interpreter, and blocks under 256 frames, are
 * done on the spot (after the queue has drained).                                                                   */
int avdsp_hip_wait_block_host(avdsp_hip_prog *prog, int max_in_flight)
{
    if (max_in_flight < 0) max_in_flight = 0;
    while ((long long)(prog->hq_submitted - prog->hq_waited) > max_in_flight) {
        auto &sl = prog->hq[prog->hq_waited % avdsp_hip_prog::kHostQueue];
        HIP_TRY(hipEventSynchronize(sl.d2h));
        if (sl.h_in)  { unpin_block(prog, sl.h_in);  sl.h_in = nullptr; }
        if (sl.h_out) { unpin_block(prog, sl.h_out); sl.h_out = nullptr; }
        prog->hq_waited++;
    }
    READY_CHECK(prog);                                   /* the blocks let through are complete: were they computed on complete windows? */
    return (int)(prog->hq_submitted - prog->hq_waited);
}

int avdsp_hip_submit_block_host(avdsp_hip_prog *prog, int plan, const void *h_in, int in_stride, int in_io_base,
                                void *h_out, int out_stride, int out_io_base, int nframes, int fir_impl, int biquad_impl)
{
    if (plan < 0 || plan >= (int)prog->plans.size()) return set_err("bad plan id %d", plan);
    Plan &pl = prog->plans[plan];
    if (pl.generic || nframes < 256 || in_stride <= 0 || out_stride <= 0) {
        if (avdsp_hip_wait_block_host(prog, 0) < 0) return -1;
        if (avdsp_hip_run_block_host(prog, plan, h_in, in_stride, in_io_base, h_out, out_stride, out_io_base, nframes, fir_impl, biquad_impl)) return -1;
        return 0;
    }
    constexpr int Q = avdsp_hip_prog::kHostQueue;
    if (avdsp_hip_wait_block_host(prog, Q - 1) < 0) return -1;            /* the slot's previous block has gone back */
    if (!prog->q_h2d) {
        HIP_TRY(hipStreamCreate(&prog->q_h2d)); HIP_TRY(hipStreamCreate(&prog->q_run)); HIP_TRY(hipStreamCreate(&prog->q_d2h));
        for (auto &sl : prog->hq)
            for (hipEvent_t *e : {&sl.h2d, &sl.run, &sl.d2h}) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    auto &sl = prog->hq[prog->hq_submitted % Q];
    const size_t in_words = (size_t)nframes * in_stride, out_words = (size_t)nframes * out_stride;
    if (sl.in_cap < in_words) {
        (void)hipFree(sl.d_in); sl.d_in = nullptr; sl.in_cap = 0;
        HIP_TRY(hipMalloc((void **)&sl.d_in, in_words * 4)); sl.in_cap = in_words;
    }
    if (sl.out_cap < out_words) {
        (void)hipFree(sl.d_out); sl.d_out = nullptr; sl.out_cap = 0;
        HIP_TRY(hipMalloc((void **)&sl.d_out, out_words * 4)); sl.out_cap = out_words;
    }
    pin_in_place(prog, h_in, in_words * 4, true);   sl.h_in = h_in;
    pin_in_place(prog, h_out, out_words * 4, true); sl.h_out = h_out;
    if (!caller_memory_is_pinned(h_in) || !caller_memory_is_pinned(h_out)) {
        /* the registration did not take (the runtime refuses some ranges): no asynchronous copy over pageable memory (copy_from_caller) --
         * the block is done on the spot, behind the queue */
        unpin_block(prog, sl.h_in); unpin_block(prog, sl.h_out); sl.h_in = sl.h_out = nullptr;
        if (avdsp_hip_wait_block_host(prog, 0) < 0) return -1;
        if (avdsp_hip_run_block_host(prog, plan, h_in, in_stride, in_io_base, h_out, out_stride, out_io_base, nframes, fir_impl, biquad_impl)) return -1;
        return 0;
    }
    const bool whole = pl.stores_whole_window && out_io_base == pl.io_out_min && out_stride == pl.io_out_max - pl.io_out_min + 1;
    if (!whole) HIP_TRY(hipMemcpyAsync(sl.d_out, h_out, out_words * 4, hipMemcpyHostToDevice, prog->q_h2d));   /* unstored slots keep their content */
    HIP_TRY(hipMemcpyAsync(sl.d_in, h_in, in_words * 4, hipMemcpyHostToDevice, prog->q_h2d));
    HIP_TRY(hipEventRecord(sl.h2d, prog->q_h2d));
    HIP_TRY(hipStreamWaitEvent(prog->q_run, sl.h2d, 0));
    prog->input_ready = sl.h2d;
    const int rc = avdsp_hip_run_block(prog, plan, sl.d_in, in_stride, in_io_base, sl.d_out, out_stride, out_io_base, nframes, fir_impl, biquad_impl, prog->q_run);
    prog->input_ready = nullptr;
    if (rc) {
        (void)hipStreamSynchronize(prog->q_h2d);
        unpin_block(prog, sl.h_in); unpin_block(prog, sl.h_out); sl.h_in = sl.h_out = nullptr;
        return -1;
    }
    HIP_TRY(hipEventRecord(sl.run, prog->q_run));
    HIP_TRY(hipStreamWaitEvent(prog->q_d2h, sl.run, 0));
    HIP_TRY(hipMemcpyAsync(h_out, sl.d_out, out_words * 4, hipMemcpyDeviceToHost, prog->q_d2h));
    HIP_TRY(hipEventRecord(sl.d2h, prog->q_d2h));
    prog->hq_submitted++;
    return (int)(prog->hq_submitted - prog->hq_waited);
}

/* Several cores over the same block.  plans[] in program order, grouped into levels (level_size[]): the host
 * has established that the cores of one level do not meet (no slot, memory word, state range or dither global
 * written by one and touched by another), so they may run at the same time; levels run one after the other.
 * Cores of a level go to side streams between a fork and a join on the caller's stream, provided every one of
 * them delivers only its own slots for this call; otherwise the level runs in order on the caller's stream. */
int avdsp_hip_run_levels(avdsp_hip_prog *prog, const int *plans, const int *level_size, int nlevels,
                         const void *d_in, int in_stride, int in_io_base, void *d_out, int out_stride, int out_io_base,
                         int nframes, int fir_impl, int biquad_impl, void *stream)
{
    hipStream_t main = (hipStream_t)stream;
    int at = 0;
    READY_CHECK(prog);
    /* windows that share IO numbers: the shared columns go from the input rows to the output rows once, here; the launches behind it
     * move their cores' slots only (show_through) */
    BlockIO wio{};
    wio.in_stride = in_stride; wio.in_base = in_io_base; wio.out_stride = out_stride; wio.out_base = out_io_base;
    wio.in = (const unsigned *)d_in; wio.out = (unsigned *)d_out; wio.nframes = nframes;
    const bool shown = windows_overlap(wio) && nframes > 1;
    if (shown && launch_show_through(wio, main)) return -1;
    ShownScope shown_scope(prog);
    if (!shown) prog->call_shown = false;
    for (int l = 0; l < nlevels; l++) {
        const int n = level_size[l];
        bool together = n > 1;
        BlockIO io{};
        io.in_stride = in_stride; io.in_base = in_io_base; io.out_stride = out_stride; io.out_base = out_io_base;
        for (int i = 0; i < n && together; i++) {
            const int id = plans[at + i];
            if (id < 0 || id >= (int)prog->plans.size()) return set_err("bad plan id %d", id);
            const Plan &pl = prog->plans[id];
            /* chain plans and interpreter cores working out of HBM share buffers; whole-window delivery overwrites */
            if (!pl.generic || !pl.ga_staged || (shown ? needs_whole(pl) : rows_whole(pl, io))) together = false;
        }
        if (!together) {
            for (int i = 0; i < n; i++)
                if (avdsp_hip_run_block(prog, plans[at + i], d_in, in_stride, in_io_base, d_out, out_stride, out_io_base,
                                        nframes, fir_impl, biquad_impl, stream)) return -1;
            at += n;
            continue;
        }
        /* all of them frame-parallel for this call: one launch, one workgroup per piece */
        if (nframes > 1 && nframes <= kFirChunk * 64) {
            constexpr int K = avdsp_hip_prog::kTableSlots;
            if (prog->table_cap < n) {
                HIP_TRY(hipDeviceSynchronize());
                (void)hipFree(prog->d_table); (void)hipHostFree(prog->h_table);
                prog->d_table = nullptr; prog->h_table = nullptr; prog->table_cap = 0;
                const int cap = std::max(n, 16);
                HIP_TRY(hipMalloc((void **)&prog->d_table, (size_t)K * cap * sizeof(GenericArgs)));
                HIP_TRY(hipHostMalloc((void **)&prog->h_table, (size_t)K * cap * sizeof(GenericArgs), hipHostMallocDefault));
                prog->table_cap = cap;
                for (auto &ev : prog->table_done) if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            }
            const int slot = prog->table_next;
            GenericArgs *table = prog->h_table + (size_t)slot * prog->table_cap;
            GenericArgs *d_slot = prog->d_table + (size_t)slot * prog->table_cap;
            HIP_TRY(hipEventSynchronize(prog->table_done[slot]));      /* (never recorded = done) */
            size_t lds = 0;
            bool grid_ok = true;
            BlockIO gio = io;
            gio.in = (const unsigned *)d_in; gio.out = (unsigned *)d_out; gio.nframes = nframes; gio.store_mask = -1;
            for (int i = 0; i < n && grid_ok; i++) {
                Plan &pl = prog->plans[plans[at + i]];
                GenericArgs a = pl.ga;
                a.io = gio;
                a.rows_whole = 0;
                if (tpdf_seq_for(prog, a, nframes)) return -1;
                if (in_stride < 0 || out_stride < 0 || in_io_base < 0 || out_io_base < 0 ||
                    (in_stride && in_io_base + in_stride > pl.ga.scratch_len) || (out_stride && out_io_base + out_stride > pl.ga.scratch_len) ||
                    pl.format != prog->plans[plans[at]].format || !wave_plan_fits(prog, pl, gio, a)) { grid_ok = false; break; }
                lds = std::max(lds, ((size_t)a.batch_lds + 128 + (size_t)a.nvm * 128 + a.seq_words) * 4);
                table[i] = a;
            }
            if (grid_ok) {
                HIP_TRY(hipMemcpyAsync(d_slot, table, (size_t)n * sizeof(GenericArgs), hipMemcpyHostToDevice, main));
                {
                    ProfileScope scope(prog, main, AVDSP_KERNEL_GENERIC_WAVE); scope.begin();
                    const dim3 grid(n), block(64);
                    switch (prog->plans[plans[at]].format) {
                    case 2:  hipLaunchKernelGGL((interp_wave_grid<2>), grid, block, lds, main, d_slot); break;
                    case 3:  hipLaunchKernelGGL((interp_wave_grid<3>), grid, block, lds, main, d_slot); break;
                    case 4:  hipLaunchKernelGGL((interp_wave_grid<4>), grid, block, lds, main, d_slot); break;
                    case 5:  hipLaunchKernelGGL((interp_wave_grid<5>), grid, block, lds, main, d_slot); break;
                    default: hipLaunchKernelGGL((interp_wave_grid<6>), grid, block, lds, main, d_slot); break;
                    }
                    HIP_TRY(hipGetLastError());
                }
                HIP_TRY(hipEventRecord(prog->table_done[slot], main));
                prog->table_next = (slot + 1) % K;
                at += n;
                continue;
            }
        }
        /* at most kSide + 1 launches in flight at a time: the cores of a level are independent of each other, so a
         * large level simply goes in several rounds */
        constexpr int kSide = 15;
        while ((int)prog->side.size() < std::min(n - 1, kSide)) {
            hipStream_t st; hipEvent_t ev;
            HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            prog->side.push_back(st);
            HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            prog->join.push_back(ev);
        }
        if (!prog->fork) HIP_TRY(hipEventCreateWithFlags(&prog->fork, hipEventDisableTiming));
        for (int c0 = 0; c0 < n; c0 += kSide + 1) {
            const int m = std::min(kSide + 1, n - c0);
            HIP_TRY(hipEventRecord(prog->fork, main));
            for (int i = 1; i < m; i++) {
                HIP_TRY(hipStreamWaitEvent(prog->side[i - 1], prog->fork, 0));
                if (avdsp_hip_run_block(prog, plans[at + c0 + i], d_in, in_stride, in_io_base, d_out, out_stride, out_io_base,
                                        nframes, fir_impl, biquad_impl, prog->side[i - 1])) return -1;
                HIP_TRY(hipEventRecord(prog->join[i - 1], prog->side[i - 1]));
            }
            if (avdsp_hip_run_block(prog, plans[at + c0], d_in, in_stride, in_io_base, d_out, out_stride, out_io_base,
                                    nframes, fir_impl, biquad_impl, stream)) return -1;
            for (int i = 1; i < m; i++) HIP_TRY(hipStreamWaitEvent(main, prog->join[i - 1], 0));
        }
        at += n;
    }
    return 0;
}

/* N instances of the program over one block each (include/avdsp_hip.h).  Every level must be frame-parallel interpreter pieces (the
 * programs this is for -- the reference's crossovers -- are); the instances' states are made on the first call after
 * avdsp_hip_set_instances as copies of the program's device state as it then is. */
/* Instances of a program made of chain cores: the mirror n times side by side in d_buf, every copy what instance 0's is now.  The
 * host then lowers each core into a plan of n x its chains (state and parameter words of instance i at i * total_words, the
 * instance's sample block as an offset in the chains' IO numbers); the kernels see a bigger plan, nothing else.  No plan may exist
 * (their rings and records address the old buffer): the host drops them first.  n <= 1: back to the one copy. */
int avdsp_hip_chain_instances(avdsp_hip_prog *prog, int n)
{
    if (n < 0 || n > 65536) return set_err("instances: 1 .. 65536");
    if (!prog->plans.empty()) return set_err("chain instances: the program's plans must be dropped first");
    if (n <= 1 && prog->chain_inst <= 1) { prog->chain_inst = 0; return 0; }
    HIP_TRY(hipDeviceSynchronize());
    const size_t W = (size_t)(prog->total_words > 0 ? prog->total_words : 1);
    const size_t Wp = AVDSP_INSTANCE_STRIDE(W);          /* (an even distance: a copy's state words keep their 8-byte alignment) */
    const size_t copies = n > 1 ? (size_t)n : 1;
    if (copies * Wp > 0x7FFFFFF0ull) return set_err("%d instances of %zu words exceed the kernels' 32-bit word indices", n, W);
    int *nb = nullptr;
    HIP_TRY(hipMalloc((void **)&nb, (copies * Wp + 2) * sizeof(int)));
    HIP_TRY(hipMemset(nb, 0, (copies * Wp + 2) * sizeof(int)));
    for (size_t i = 0; i < copies; i++)
        if (hipMemcpy(nb + i * Wp, prog->d_buf, W * sizeof(int), hipMemcpyDeviceToDevice) != hipSuccess) { (void)hipFree(nb); return set_err("chain instances: copying the mirror failed"); }
    (void)hipFree(prog->d_buf);
    prog->d_buf = nb;
    prog->chain_inst = n > 1 ? n : 0;
    return 0;
}

int avdsp_hip_set_instances(avdsp_hip_prog *prog, int n)
{
    if (n < 1 || n > 65536) return set_err("instances: 1 .. 65536");
    HIP_TRY(hipDeviceSynchronize());
    prog->inst_n = n; prog->inst_valid = false;
    return 0;
}

static int instances_ready(avdsp_hip_prog *prog, int nframes)
{
    const int n = prog->inst_n;
    const size_t bw = (size_t)prog->total_words + 2;
    const int seqf = std::max(prog->tpdf_seq_frames, std::max(nframes, 4096));
    if (prog->inst_valid && prog->inst_frame_words == prog->frame_words && prog->inst_seq_frames >= nframes) return 0;
    HIP_TRY(hipDeviceSynchronize());
    const bool fresh = !prog->inst_valid;
    if (fresh || prog->inst_frame_words != prog->frame_words) {
        if (!fresh) return set_err("instances: the program's frame grew after the instances were made (run every core once before dspRuntimeSetInstances, or set them again)");
        (void)hipFree(prog->d_inst_buf); (void)hipFree(prog->d_inst_tpdf); (void)hipFree(prog->d_inst_frame);
        prog->d_inst_buf = nullptr; prog->d_inst_tpdf = nullptr; prog->d_inst_frame = nullptr;
        if (n > 1) {
            if (!prog->d_tpdf) return set_err("instances: the program has no dither state (dspRuntimeReset first)");
            HIP_TRY(hipMalloc((void **)&prog->d_inst_buf, (size_t)(n - 1) * bw * sizeof(int)));
            HIP_TRY(hipMalloc((void **)&prog->d_inst_tpdf, (size_t)(n - 1) * sizeof(TpdfGlobals)));
            HIP_TRY(hipMalloc((void **)&prog->d_inst_frame, (size_t)(n - 1) * std::max(prog->frame_words, 1) * sizeof(unsigned)));
            for (int i = 0; i < n - 1; i++) {
                HIP_TRY(hipMemcpyAsync(prog->d_inst_buf + (size_t)i * bw, prog->d_buf, bw * sizeof(int), hipMemcpyDeviceToDevice, nullptr));
                HIP_TRY(hipMemcpyAsync(prog->d_inst_tpdf + i, prog->d_tpdf, sizeof(TpdfGlobals), hipMemcpyDeviceToDevice, nullptr));
                if (prog->frame_words)
                    HIP_TRY(hipMemcpyAsync(prog->d_inst_frame + (size_t)i * prog->frame_words, prog->d_frame, (size_t)prog->frame_words * 4, hipMemcpyDeviceToDevice, nullptr));
            }
        }
        prog->inst_frame_words = prog->frame_words;
        prog->inst_seq_frames = 0;
    }
    if (prog->inst_seq_frames < nframes) {
        (void)hipFree(prog->d_inst_seq); prog->d_inst_seq = nullptr;
        if (n > 1) {
            HIP_TRY(hipMalloc((void **)&prog->d_inst_seq, (size_t)(n - 1) * seqf * 2 * sizeof(int)));
            HIP_TRY(hipMemset(prog->d_inst_seq, 0, (size_t)(n - 1) * seqf * 2 * sizeof(int)));
        }
        prog->inst_seq_frames = seqf;
    }
    HIP_TRY(hipDeviceSynchronize());
    prog->inst_valid = true;
    return 0;
}

int avdsp_hip_run_levels_instances(avdsp_hip_prog *prog, const int *plans, const int *level_size, int nlevels,
                                   const void *d_in, int in_stride, int in_io_base, size_t in_inst_words,
                                   void *d_out, int out_stride, int out_io_base, size_t out_inst_words, int nframes, void *stream)
{
    hipStream_t main = (hipStream_t)stream;
    const int ninst = prog->inst_n;
    if (nframes < 2 || nframes > kFirChunk * 64) return set_err("instances: blocks of 2 .. %d frames", kFirChunk * 64);
    /* (the dither-pair buffers and the frame exist once the pieces have been looked at: a first pass over the table) */
    constexpr int K = avdsp_hip_prog::kTableSlots;
    BlockIO gio{};
    gio.in_stride = in_stride; gio.in_base = in_io_base; gio.out_stride = out_stride; gio.out_base = out_io_base;
    gio.in = (const unsigned *)d_in; gio.out = (unsigned *)d_out; gio.nframes = nframes; gio.store_mask = -1;
    if (windows_overlap(gio) && launch_show_through(gio, main, ninst, in_inst_words, out_inst_words)) return -1;     /* (the callers' rows: nothing of the instances' state) */
    int at = 0;
    for (int l = 0; l < nlevels; l++) {
        const int nl = level_size[l];
        /* The pieces of a level go out as ONE grid of pieces x instances -- unless a piece delivers whole rows (the call's windows share IO
         * numbers, or the host could not tell what the core owns): whole rows of two pieces would overwrite each other, so such a
         * level's pieces run one after the other, each as a grid of its own over the instances (avdsp_hip_run_levels does the same
         * with one launch per piece). */
        bool together = true;
        for (int i = 0; i < nl; i++) {
            const int id = plans[at + i];
            if (id < 0 || id >= (int)prog->plans.size()) return set_err("bad plan id %d", id);
            const Plan &pl = prog->plans[id];
            if (!pl.generic || !pl.ga_staged)
                return set_err("instances: every core must run on the frame-parallel interpreter (plan %d does not)", id);
            if (needs_whole(pl)) together = false;          /* (windows that share IO numbers: their columns were copied in front, show_through) */
        }
        for (int first = 0; first < nl; first += together ? nl : 1) {
            const int n = together ? nl : 1;
            if (prog->table_cap < n) {
                HIP_TRY(hipDeviceSynchronize());
                (void)hipFree(prog->d_table); (void)hipHostFree(prog->h_table);
                prog->d_table = nullptr; prog->h_table = nullptr; prog->table_cap = 0;
                const int cap = std::max(n, 16);
                HIP_TRY(hipMalloc((void **)&prog->d_table, (size_t)K * cap * sizeof(GenericArgs)));
                HIP_TRY(hipHostMalloc((void **)&prog->h_table, (size_t)K * cap * sizeof(GenericArgs), hipHostMallocDefault));
                prog->table_cap = cap;
                for (auto &ev : prog->table_done) if (!ev) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            }
            const int slot = prog->table_next;
            GenericArgs *table = prog->h_table + (size_t)slot * prog->table_cap;
            GenericArgs *d_slot = prog->d_table + (size_t)slot * prog->table_cap;
            HIP_TRY(hipEventSynchronize(prog->table_done[slot]));
            size_t lds = 0;
            int fmt = 0;
            for (int i = 0; i < n; i++) {
                const int id = plans[at + first + i];
                Plan &pl = prog->plans[id];
                GenericArgs a = pl.ga;
                a.io = gio;
                a.rows_whole = together ? 0 : needs_whole(pl);
                if (tpdf_seq_for(prog, a, nframes)) return -1;
                if (in_stride < 0 || out_stride < 0 || in_io_base < 0 || out_io_base < 0 ||
                    (in_stride && in_io_base + in_stride > pl.ga.scratch_len) || (out_stride && out_io_base + out_stride > pl.ga.scratch_len) ||
                    (fmt && pl.format != fmt) || !wave_plan_fits(prog, pl, gio, a))
                    return set_err("instances: plan %d does not fit the frame-parallel interpreter for these windows", id);
                fmt = pl.format;
                lds = std::max(lds, ((size_t)a.batch_lds + 128 + (size_t)a.nvm * 128 + a.seq_words) * 4);
                table[i] = a;
            }
            if (instances_ready(prog, nframes)) return -1;
            InstanceStrides st{};
            st.n = n; st.buf = prog->d_inst_buf; st.buf_words = (size_t)prog->total_words + 2; st.tpdf = prog->d_inst_tpdf;
            st.scratch = prog->d_inst_frame; st.frame_words = (size_t)prog->frame_words;
            st.tpdf_seq = prog->d_inst_seq; st.seq_words = (size_t)prog->inst_seq_frames * 2;
            st.in_words = in_inst_words; st.out_words = out_inst_words;
            HIP_TRY(hipMemcpyAsync(d_slot, table, (size_t)n * sizeof(GenericArgs), hipMemcpyHostToDevice, main));
            {
                ProfileScope scope(prog, main, AVDSP_KERNEL_GENERIC_WAVE); scope.begin();
                const dim3 grid((unsigned)n * (unsigned)ninst), block(64);
                switch (fmt) {
                case 2:  hipLaunchKernelGGL((interp_wave_instances<2>), grid, block, lds, main, d_slot, st); break;
                case 3:  hipLaunchKernelGGL((interp_wave_instances<3>), grid, block, lds, main, d_slot, st); break;
                case 4:  hipLaunchKernelGGL((interp_wave_instances<4>), grid, block, lds, main, d_slot, st); break;
                case 5:  hipLaunchKernelGGL((interp_wave_instances<5>), grid, block, lds, main, d_slot, st); break;
                default: hipLaunchKernelGGL((interp_wave_instances<6>), grid, block, lds, main, d_slot, st); break;
                }
                HIP_TRY(hipGetLastError());
            }
            HIP_TRY(hipEventRecord(prog->table_done[slot], main));
            prog->table_next = (slot + 1) % K;
        }
        at += nl;
    }
    return 0;
}

/* the data area (or any word range) of one instance's mirror, for the host */
int avdsp_hip_download_instance_words(avdsp_hip_prog *p, int inst, int32_t *host_buf, int first, int n)
{
    if (inst < 0 || inst >= std::max(p->inst_n, p->chain_inst)) return set_err("instance %d of %d", inst, std::max(p->inst_n, p->chain_inst));
    if (check_range(p, first, n)) return -1;
    HIP_TRY(hipDeviceSynchronize());
    if (p->chain_inst > 1) {                             /* chain instances: copy `inst` of the mirror, the FIR histories brought home first */
        if (inst >= p->chain_inst) return set_err("instance %d of %d", inst, p->chain_inst);
        READY_CHECK(p);
        if (rings_to_mirror(p)) return -1;
        HIP_TRY(hipDeviceSynchronize());
        if (copy_to_caller(host_buf, p->d_buf + (size_t)inst * AVDSP_INSTANCE_STRIDE(p->total_words) + first, (size_t)n * sizeof(int))) return -1;
        return 0;
    }
    /* (instance 0 of a program with ONE instance may be a chain program like any other: its FIR histories live in the plans' rings) */
    READY_CHECK(p);
    if (rings_to_mirror(p)) return -1;
    HIP_TRY(hipDeviceSynchronize());
    const int *src = p->d_buf;
    if (inst > 0) {
        if (!p->inst_valid || !p->d_inst_buf) return set_err("the instances have not run yet");
        src = p->d_inst_buf + (size_t)(inst - 1) * ((size_t)p->total_words + 2);
    }
    if (copy_to_caller(host_buf, src + first, (size_t)n * sizeof(int))) return -1;
    return 0;
}

int avdsp_hip_run_levels_host(avdsp_hip_prog *prog, const int *plans, const int *level_size, int nlevels,
                              const void *h_in, int in_stride, int in_io_base, void *h_out, int out_stride, int out_io_base,
                              int nframes, int fir_impl, int biquad_impl)
{
    const size_t in_words = (size_t)nframes * in_stride, out_words = (size_t)nframes * out_stride;
    if (prog->in_cap < in_words) {
        (void)hipFree(prog->d_in); prog->d_in = nullptr; prog->in_cap = 0;
        HIP_TRY(hipMalloc((void **)&prog->d_in, in_words * 4)); prog->in_cap = in_words;
    }
    if (prog->out_cap < out_words) {
        (void)hipFree(prog->d_out); prog->d_out = nullptr; prog->out_cap = 0;
        HIP_TRY(hipMalloc((void **)&prog->d_out, out_words * 4)); prog->out_cap = out_words;
    }
    if (copy_from_caller(prog->d_in, h_in, in_words * 4)) return -1;
    if (copy_from_caller(prog->d_out, h_out, out_words * 4)) return -1;   /* unstored slots keep their content */
    if (avdsp_hip_run_levels(prog, plans, level_size, nlevels, prog->d_in, in_stride, in_io_base, prog->d_out, out_stride,
                             out_io_base, nframes, fir_impl, biquad_impl, nullptr)) return -1;
    HIP_TRY(hipDeviceSynchronize());
    if (copy_to_caller(h_out, prog->d_out, out_words * 4)) return -1;
    return 0;
}

int avdsp_hip_unpack_pcm(avdsp_hip_prog *prog, int pcm, const void *d_src, void *d_dst, size_t nsamples, void *stream)
{
    if (pcm != AVDSP_PCM_S24_3LE && pcm != AVDSP_PCM_S16) return set_err("unpack: PCM kind %d needs no conversion or is unknown", pcm);
    if (reinterpret_cast<size_t>(d_dst) & 15) return set_err("unpack: destination must be 16-byte aligned");
    if (!nsamples) return 0;
    ProfileScope scope(prog, (hipStream_t)stream, AVDSP_KERNEL_UNPACK); scope.begin();
    UnpackArgs a{(const unsigned char *)d_src, (unsigned *)d_dst, nsamples};
    const int grid = (int)std::min<size_t>((nsamples / 4 + kBlock - 1) / kBlock + 1, 8192);
    if (pcm == AVDSP_PCM_S24_3LE) hipLaunchKernelGGL(pcm_unpack<AVDSP_PCM_S24_3LE>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, a);
    else                          hipLaunchKernelGGL(pcm_unpack<AVDSP_PCM_S16>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

/* the block that follows reads what pcm_unpack (null stream) is still writing: the overlap mode's cascade, on its own
 * non-blocking stream, waits for this event (launch_all) */
static int unpack_done(avdsp_hip_prog *prog)
{
    if (!prog->ev_unpack) HIP_TRY(hipEventCreateWithFlags(&prog->ev_unpack, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(prog->ev_unpack, nullptr));
    prog->input_ready = prog->ev_unpack;
    return 0;
}

int avdsp_hip_run_block_pcm_host(avdsp_hip_prog *prog, int plan, int pcm, const void *h_src, int in_stride, int in_io_base,
                                 void *h_out, int out_stride, int out_io_base, int nframes,
                                 int fir_impl, int biquad_impl)
{
    if (plan < 0 || plan >= (int)prog->plans.size()) return set_err("bad plan id %d", plan);
    if (pcm == AVDSP_PCM_S32)
        return avdsp_hip_run_block_host(prog, plan, h_src, in_stride, in_io_base, h_out, out_stride, out_io_base, nframes, fir_impl, biquad_impl);
    if (pcm != AVDSP_PCM_S24_3LE && pcm != AVDSP_PCM_S16) return set_err("unknown PCM kind %d", pcm);
    if (prog->plans[plan].format > 4) return set_err("packed PCM input feeds the int-sample formats 2, 3, 4 only");
    if (nframes <= 0) return 0;
    const size_t nsamples = (size_t)nframes * in_stride, out_words = (size_t)nframes * out_stride;
    const size_t raw_bytes = nsamples * (pcm == AVDSP_PCM_S16 ? 2 : 3), raw_words = (raw_bytes + 3) / 4;
    /* staging: [unpacked samples | raw bytes] in d_in, output rows in d_out */
    const size_t unpacked = (nsamples + 3) & ~(size_t)3;
    if (prog->in_cap < unpacked + raw_words) {
        (void)hipFree(prog->d_in); prog->d_in = nullptr; prog->in_cap = 0;
        HIP_TRY(hipMalloc((void **)&prog->d_in, (unpacked + raw_words) * 4)); prog->in_cap = unpacked + raw_words;
    }
    if (prog->out_cap < out_words) {
        (void)hipFree(prog->d_out); prog->d_out = nullptr; prog->out_cap = 0;
        HIP_TRY(hipMalloc((void **)&prog->d_out, out_words * 4)); prog->out_cap = out_words;
    }
    unsigned *d_raw = prog->d_in + unpacked;
    if (copy_from_caller(d_raw, h_src, raw_bytes)) return -1;
    if (copy_from_caller(prog->d_out, h_out, out_words * 4)) return -1;
    if (avdsp_hip_unpack_pcm(prog, pcm, d_raw, prog->d_in, nsamples, nullptr) || unpack_done(prog)) return -1;
    const int rc = avdsp_hip_run_block(prog, plan, prog->d_in, in_stride, in_io_base, prog->d_out, out_stride, out_io_base,
                                       nframes, fir_impl, biquad_impl, nullptr);
    prog->input_ready = nullptr;
    if (rc) return -1;
    HIP_TRY(hipDeviceSynchronize());
    if (copy_to_caller(h_out, prog->d_out, out_words * 4)) return -1;
    return 0;
}

/* packed PCM in front of avdsp_hip_run_levels_host: the block is uploaded and unpacked once for all cores */
int avdsp_hip_run_levels_pcm_host(avdsp_hip_prog *prog, const int *plans, const int *level_size, int nlevels, int pcm,
                                  const void *h_src, int in_stride, int in_io_base, void *h_out, int out_stride, int out_io_base,
                                  int nframes, int fir_impl, int biquad_impl)
{
    if (pcm == AVDSP_PCM_S32)
        return avdsp_hip_run_levels_host(prog, plans, level_size, nlevels, h_src, in_stride, in_io_base, h_out, out_stride,
                                         out_io_base, nframes, fir_impl, biquad_impl);
    if (pcm != AVDSP_PCM_S24_3LE && pcm != AVDSP_PCM_S16) return set_err("unknown PCM kind %d", pcm);
    int total = 0;
    for (int l = 0; l < nlevels; l++) total += level_size[l];
    for (int i = 0; i < total; i++) {
        if (plans[i] < 0 || plans[i] >= (int)prog->plans.size()) return set_err("bad plan id %d", plans[i]);
        if (prog->plans[plans[i]].format > 4) return set_err("packed PCM input feeds the int-sample formats 2, 3, 4 only");
    }
    if (nframes <= 0) return 0;
    const size_t nsamples = (size_t)nframes * in_stride, out_words = (size_t)nframes * out_stride;
    const size_t raw_bytes = nsamples * (pcm == AVDSP_PCM_S16 ? 2 : 3), raw_words = (raw_bytes + 3) / 4;
    const size_t unpacked = (nsamples + 3) & ~(size_t)3;          /* staging: [unpacked samples | raw bytes] in d_in */
    if (prog->in_cap < unpacked + raw_words) {
        (void)hipFree(prog->d_in); prog->d_in = nullptr; prog->in_cap = 0;
        HIP_TRY(hipMalloc((void **)&prog->d_in, (unpacked + raw_words) * 4)); prog->in_cap = unpacked + raw_words;
    }
    if (prog->out_cap < out_words) {
        (void)hipFree(prog->d_out); prog->d_out = nullptr; prog->out_cap = 0;
        HIP_TRY(hipMalloc((void **)&prog->d_out, out_words * 4)); prog->out_cap = out_words;
    }
    unsigned *d_raw = prog->d_in + unpacked;
    if (copy_from_caller(d_raw, h_src, raw_bytes)) return -1;
    if (copy_from_caller(prog->d_out, h_out, out_words * 4)) return -1;
    if (avdsp_hip_unpack_pcm(prog, pcm, d_raw, prog->d_in, nsamples, nullptr) || unpack_done(prog)) return -1;
    const int rc = avdsp_hip_run_levels(prog, plans, level_size, nlevels, prog->d_in, in_stride, in_io_base, prog->d_out, out_stride,
                                        out_io_base, nframes, fir_impl, biquad_impl, nullptr);
    prog->input_ready = nullptr;
    if (rc) return -1;
    HIP_TRY(hipDeviceSynchronize());
    if (copy_to_caller(h_out, prog->d_out, out_words * 4)) return -1;
    return 0;
}

int avdsp_hip_tag_output(avdsp_hip_prog *prog, void *d_column, int stride, int nframes, int reset, int reset_value, void *stream)
{
    if (!prog->d_tag_prev) {
        HIP_TRY(hipMalloc((void **)&prog->d_tag_prev, sizeof(int)));
        HIP_TRY(hipMemset(prog->d_tag_prev, 0, sizeof(int)));
    }
    if (reset) HIP_TRY(hipMemcpyAsync(prog->d_tag_prev, &reset_value, sizeof(int), hipMemcpyHostToDevice, (hipStream_t)stream));
    if (reset) HIP_TRY(hipStreamSynchronize((hipStream_t)stream));           /* reset_value lives on the caller's stack */
    if (nframes <= 0) return 0;
    TagArgs a{(int *)d_column, stride, nframes, prog->d_tag_prev};
    hipLaunchKernelGGL(tag_column, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int avdsp_hip_tag_column_host(avdsp_hip_prog *prog, int *h_column, int nframes)
{
    int *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, (size_t)nframes * sizeof(int)));
    hipError_t e = hipMemcpy(d, h_column, (size_t)nframes * sizeof(int), hipMemcpyHostToDevice);
    int rc = e == hipSuccess ? avdsp_hip_tag_output(prog, d, 1, nframes, 0, 0, nullptr) : set_err("hipMemcpy: %s", hipGetErrorString(e));
    if (!rc) { e = hipMemcpy(h_column, d, (size_t)nframes * sizeof(int), hipMemcpyDeviceToHost); if (e != hipSuccess) rc = set_err("hipMemcpy: %s", hipGetErrorString(e)); }
    (void)hipFree(d);
    return rc;
}

int avdsp_hip_prog_set_option(avdsp_hip_prog *prog, int key, int value)
{
    HIP_TRY(hipDeviceSynchronize());                    /* nothing in flight when the launch arrangement changes */
    switch (key) {
    case AVDSP_OPT_OVERLAP:  prog->overlap = value; for (bool &f : prog->ev_fir_set) f = false; return 0;
    case AVDSP_OPT_READY_WORDS: if (value < -1 || value > 2) return set_err("ready_words: -1 (by plan), 0 (events), 1 (the cascade's waves publish) or 2 (a kernel behind the cascade publishes)"); prog->ready_words = value; return 0;
    case AVDSP_OPT_LANE_HW: prog->lane_hw = value != 0; return 0;
    case AVDSP_OPT_FIR_SPLIT: prog->fir_split = value != 0; return 0;
    case AVDSP_OPT_FIR_LEAN: if (value < -1 || value > 1) return set_err("fir_lean: -1 (auto), 0 or 1"); prog->fir_lean = value; return 0;
    case AVDSP_OPT_RING_WAIT: prog->ring_wait_host = value != 0; return 0;
    case AVDSP_OPT_READY_TEST: prog->ready_test = value > 0 ? value : 0; return 0;
    case AVDSP_OPT_GROUP_FANOUT: prog->group_fanout = value != 0; return 0;
    case AVDSP_OPT_CU_SPLIT:
        if (value < -128 || value > 128 || (value & 7)) return set_err("cu_split: 0 (off) or 8, 16, ... 128 CUs for the cascades' stream (negative: the FIRs stay on the caller's stream, unmasked)");
        if (value != prog->cu_split) {                   /* the side streams are made anew with their masks at the next overlapped launch */
            if (prog->s_bq) { (void)hipStreamDestroy(prog->s_bq); prog->s_bq = nullptr; }
            for (auto &fs : prog->s_fir) if (fs) { (void)hipStreamDestroy(fs); fs = nullptr; }
            for (int i = 0; i < avdsp_hip_prog::kAhead; i++) {
                if (prog->ev_bq[i]) { (void)hipEventDestroy(prog->ev_bq[i]); prog->ev_bq[i] = nullptr; }
                if (prog->ev_fir[i]) { (void)hipEventDestroy(prog->ev_fir[i]); prog->ev_fir[i] = nullptr; }
                prog->ev_fir_set[i] = false; prog->ev_fir_now[i] = nullptr;
            }
            prog->probed.clear();
            prog->cu_split = value;
        }
        return 0;
    case AVDSP_OPT_FIR_LAUNCH: if (value < -1 || value > 2) return set_err("fir_launch: -1 (auto), 0, 1 or 2"); prog->fir_launch_mode = value; return 0;
    case AVDSP_OPT_FIR_ROWS: if (value != 0 && value != 1 && value != 2 && value != 4) return set_err("fir_tile row tiles: 0 (auto), 1, 2 or 4");
                             prog->fir_rows = value; return 0;
    case AVDSP_OPT_PROFILE_STRIDE: if (value < 1) return set_err("profile_stride: every n-th launch, n >= 1"); prog->profile_stride = value; return 0;
    case AVDSP_OPT_HOST_SPLIT: if (value < 0) return set_err("host_split: frames per piece, 0 = whole block"); prog->host_split = value; return 0;
    case AVDSP_OPT_HOST_PIN: prog->host_pin = value != 0;
                             if (!value) {           /* (nothing is in flight: set_option synchronised above; queued blocks keep their references until waited for) */
                                 (void)avdsp_hip_wait_block_host(prog, 0);
                                 for (auto &pn : prog->pinned) if (pn.ours) (void)hipHostUnregister(const_cast<void *>(pn.ptr));
                                 prog->pinned.clear();
                             }
                             return 0;
    }
    return set_err("unknown device option %d", key);
}

/* what the launch arrangement turned out to be (dspRuntimeGetOption "side_by_side", "ready_mode") */
int avdsp_hip_prog_get_option(avdsp_hip_prog *prog, int key)
{
    switch (key) {
    case AVDSP_OPT_SIDE_BY_SIDE: return prog->s_bq ? prog->side_by_side : -1;      /* -1: not probed yet (no overlapped launch so far) */
    case AVDSP_OPT_READY_MODE:   return prog->ready_mode_now;
    case AVDSP_OPT_STREAMS_REMADE: return prog->remade;
    }
    return -1;
}

/* the caller has heard of the time-outs (dspRuntimeReset; dspRuntimeSetOption("ready_timeouts", 0)): count and mark start again */
int avdsp_hip_ready_clear(avdsp_hip_prog *prog)
{
    if (!prog->d_ready_timeouts) return 0;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(prog->d_ready_timeouts, 0, 4));
    if (prog->h_ready_flag) *(volatile unsigned *)prog->h_ready_flag = 0;
    return 0;
}

int avdsp_hip_last_error_is_ready_timeout(void) { return g_err_ready ? 1 : 0; }

int avdsp_hip_ready_timeouts(avdsp_hip_prog *prog)
{
    unsigned n = 0;
    if (!prog->d_ready_timeouts) return 0;
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(&n, prog->d_ready_timeouts, sizeof n, hipMemcpyDeviceToHost));
    return (int)std::min(n, 0x7FFFFFFFu);
}

int avdsp_hip_profile_enable(avdsp_hip_prog *prog, int on)
{
    prog->profile = on == 1 ? ~0u : (unsigned)on >> 1;     /* 0 off, 1 every kind, otherwise 2 * (mask of kinds) */
    return 0;
}

int avdsp_hip_profile_read(avdsp_hip_prog *prog, int kind, double *total_ms, int *launches)
{
    double sum = 0.0;
    int n = 0, pairs = 0;
    std::vector<avdsp_hip_prog::Span> keep;
    for (auto &sp : prog->spans) {
        if (sp.kind != kind) { keep.push_back(sp); continue; }
        HIP_TRY(hipEventSynchronize(sp.b));
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, sp.a, sp.b));
        sum += ms; n++; pairs += sp.pair;
        for (int i = 0; i < avdsp_hip_prog::kAhead; i++)       /* (an event that stood for a FIR's end goes back to the pool: that FIR HAS ended) */
            if (prog->ev_fir_now[i] == sp.b) { prog->ev_fir_now[i] = prog->ev_fir[i]; prog->ev_fir_set[i] = false; }
        prog->free_events.push_back(sp.a); prog->free_events.push_back(sp.b);
    }
    prog->last_read_pairs[kind & 7] = pairs;
    prog->spans.swap(keep);
    if (total_ms) *total_ms = sum;
    if (launches) *launches = n;
    return 0;
}

int avdsp_hip_profile_last_pairs(avdsp_hip_prog *prog, int kind) { return prog->last_read_pairs[kind & 7]; }

int avdsp_hip_synchronize(void *stream)
{
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

}  // extern "C"
