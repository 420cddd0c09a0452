/*
 * avdsp_oracle.c -- TEST INFRASTRUCTURE (see avdsp_oracle.h).  CPU restatement of the reference
 * AVDSP runtime, written from the behaviour of module_avdsp/runtime (c and h files); every block cites
 * the lines it follows.  Not linked into, loaded by, or called from the product path.
 */
#include "avdsp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * bit-level views
 * ---------------------------------------------------------------------------------------- */
static inline uint32_t f32_bits(float f)   { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float    f32_from(uint32_t u){ float f;    memcpy(&f, &u, 4); return f; }
static inline uint64_t f64_bits(double d)  { uint64_t u; memcpy(&u, &d, 8); return u; }
static inline double   f64_from(uint64_t u){ double d;   memcpy(&d, &u, 8); return d; }

/* 64-bit state words live at 32-bit granularity inside the caller's int buffer */
static inline int64_t ld_i64(const int *p)          { int64_t v; memcpy(&v, p, 8); return v; }
static inline void    st_i64(int *p, int64_t v)     { memcpy(p, &v, 8); }
static inline double  ld_f64(const int *p)          { double v; memcpy(&v, p, 8); return v; }
static inline void    st_f64(int *p, double v)      { memcpy(p, &v, 8); }
static inline float   ld_f32(const int *p)          { float v; memcpy(&v, p, 4); return v; }
static inline void    st_f32(int *p, float v)       { memcpy(p, &v, 4); }

/* wrapping signed arithmetic (the reference relies on two's complement wrap under -Ofast) */
static inline int64_t wadd(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }
static inline int64_t wsub(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }
static inline int64_t wmul(int64_t a, int64_t b) { return (int64_t)((uint64_t)a * (uint64_t)b); }
static inline int64_t wshl(int64_t a, int n)     { return (int64_t)((uint64_t)a << (n & 63)); }
static inline int64_t ashr(int64_t a, int n)     { return a >> (n & 63); }

/* ------------------------------------------------------------------------------------------
 * dsp_ieee754.h restated.  The reference builds with DSP_IEEE754_OPTIMISE = 63
 * (dsp_runtime.c:10), i.e. every bit-level fast path is the one in effect.
 * ---------------------------------------------------------------------------------------- */

/* dsp_ieee754.h:377-410 -- exact product of two floats as a double; an operand whose biased
 * exponent is 0 (zero or subnormal) gives +0.0; Inf/NaN are not recognised.                  */
double oracle_mul_float_double(float a, float b)
{
    uint32_t ua = f32_bits(a), ub = f32_bits(b);
    int ea = (ua >> 23) & 255, eb = (ub >> 23) & 255;
    if (ea == 0 || eb == 0) return 0.0;
    int e = 1023 + ea + eb - 254;
    if (e < 1) return 0.0;                       /* unreachable for ea,eb >= 1; kept for fidelity */
    uint64_t ma = (ua & 0x7FFFFFu) | 0x800000u, mb = (ub & 0x7FFFFFu) | 0x800000u;
    uint64_t m = ma * mb;                         /* 47 or 48 significant bits */
    if (m & 0x800000000000ull) { e++; m <<= 5; } else m <<= 6;
    m &= (1ull << 52) - 1;
    uint64_t r = m | ((uint64_t)(unsigned)e << 52);
    if ((ua ^ ub) & 0x80000000u) r |= 1ull << 63;
    return f64_from(r);
}

/* dsp_ieee754.h:342-375 -- float product with the mantissa TRUNCATED (not rounded) to 24 bits */
float oracle_mul_float_float(float a, float b)
{
    uint32_t ua = f32_bits(a), ub = f32_bits(b);
    int ea = (ua >> 23) & 255, eb = (ub >> 23) & 255;
    if (ea == 0 || eb == 0) return 0.0f;
    int e = ea + eb - 127;
    if (e < 1) return 0.0f;
    if ((ua ^ ub) & 0x80000000u) e |= 1 << 8;
    uint64_t ma = ((ua & 0x7FFFFFu) | 0x800000u) << 5, mb = ((ub & 0x7FFFFFu) | 0x800000u) << 5;
    uint32_t hi = (uint32_t)((ma * mb) >> 32);
    if (hi & (1u << 25)) { e++; hi >>= 2; } else hi >>= 1;
    hi &= (1u << 23) - 1;
    hi |= (uint32_t)e << 23;
    return f32_from(hi);
}

/* acc += a*b for a float accumulator (dspMaccFloatFloat, dsp_ieee754.h:412-417).  When the product takes
 * one of dspMulFloatFloat's "return 0.0" exits, the reference's build does not add at all (-Ofast implies
 * -fno-signed-zeros, x + 0.0 is x): an accumulator holding -0.0 -- a flushed negative underflow -- stays
 * -0.0 where an IEEE addition of +0.0 would turn it into +0.0. */
static inline float macc_float_float(float acc, float a, float b)
{
    uint32_t ua = f32_bits(a), ub = f32_bits(b);
    int ea = (ua >> 23) & 255, eb = (ub >> 23) & 255;
    if (ea == 0 || eb == 0 || ea + eb - 127 < 1) return acc;
    return acc + oracle_mul_float_float(a, b);
}

/* dsp_ieee754.h:204-250 -- int -> float with the magnitude truncated to 24 bits, scaled by
 * 2^-shift.  INT_MIN takes the reference's 7-step normalisation one step short and comes out
 * as mantissa 0 with exponent 157-shift (i.e. -0.5 for shift 31).                            */
float oracle_int_to_float_scaled(int x, int shift)
{
    if (x == 0) return 0.0f;
    uint32_t sign = 0, mag = (uint32_t)x;
    if (x < 0) { sign = 256; mag = 0u - mag; }
    int e;
    if (mag == 0x80000000u) { e = 157; mag = 0; }
    else {
        int p = 31 - __builtin_clz(mag);          /* index of the leading one */
        if (p > 23) mag >>= (p - 23); else mag <<= (23 - p);
        e = 127 + p;
    }
    e -= shift;
    uint32_t r = (mag & 0x7FFFFFu) | ((uint32_t)(e + (int)sign) << 23);
    return f32_from(r);
}

/* dsp_ieee754.h:253-298 -- exact: every int32 fits a double mantissa, so the bit assembly there
 * equals x * 2^-shift.  The one exception is INT_MIN, where the reference negates an int (signed
 * overflow, undefined); its gcc -Ofast build returns -2^(23-shift), i.e. -2^-8 for shift 31, and
 * that observed value is mirrored here (golden vector kernel_vectors.npz, iv[4]).            */
double oracle_int_to_double_scaled(int x, int shift)
{
    if (x == 0) return 0.0;
    if (x == (int)0x80000000u) return -ldexp(1.0, 23 - shift);
    return ldexp((double)x, -shift);
}

/* dsp_ieee754.h:60-83 */
int oracle_s31_from_float(float f)
{
    uint32_t u = f32_bits(f);
    int e = (u >> 23) & 255;
    if (e == 0) return 0;
    uint32_t m = ((u & 0x7FFFFFu) | 0x800000u) << 8;
    int n = 127 - e;
    /* |f| < 2^-32: the reference shifts a 32-bit value by >= 32, undefined in C; its binaries on x86-64
     * (and AArch64) take the count modulo 32, which is what the golden vectors contain */
    if (n > 0) m >>= (n & 31);
    else m = 0x7FFFFFFFu;
    if (u & 0x80000000u) m = 0u - m;
    return (int)m;
}

/* dsp_ieee754.h:85-107 -- truncation toward zero; |d| >= 1 gives +/-0x7FFFFFFF; exponent 0 gives 0.
 * For |d| < 2^-42 the reference shifts a 64-bit value by >= 64: undefined in C, count modulo 64 in its
 * x86-64 / AArch64 binaries, and therefore here. */
int oracle_s31_from_double(double d)
{
    uint64_t u = f64_bits(d);
    int e = (int)((u >> 52) & 2047);
    if (e == 0) return 0;
    int64_t m = (int64_t)((u & ((1ull << 52) - 1)) | (1ull << 52));
    int n = 1044 - e;
    if (n > 21) m >>= (n & 63);
    else m = 0x7FFFFFFF;
    if ((int64_t)u < 0) m = -m;
    return (int)m;
}

/* dsp_ieee754.h:171-184 */
float oracle_saturate_float(float f)
{
    int e = (int)f32_bits(f) >> 23;               /* arithmetic: sign kept */
    if (e >= 127) return 1.0f;
    if (e < 0 && e >= -129) return -1.0f;
    return f;
}

/* dsp_ieee754.h:187-199 */
double oracle_saturate_double(double d)
{
    int e = (int)((int64_t)f64_bits(d) >> 52);
    if (e >= 1023) return 1.0;
    if (e < 0 && e >= -1025) return -1.0;
    return d;
}

/* dsp_ieee754.h:112-139 -- keep `bit` bits of a value in -1..1 (floor toward -inf) */
float oracle_truncate_float(float f, int bit)
{
    int32_t i = (int32_t)f32_bits(f);
    int e = (i >> 23) & 255;
    if (e == 0) return 0.0f;
    int n = 151 - bit - e;
    if (n > 0) {
        if (n >= 24) i = (i >= 0) ? 0 : (int32_t)((uint32_t)(256 + 128 - bit) << 23);
        else {
            int32_t mask = (int32_t)(0xFFFFFFFFu << n);
            if (i < 0) i = (int32_t)((uint32_t)i + (uint32_t)~mask);
            i &= mask;
        }
    }
    return f32_from((uint32_t)i);
}

/* dsp_ieee754.h:142-168 */
double oracle_truncate_double(double d, int bit)
{
    int64_t i = (int64_t)f64_bits(d);
    int e = (int)((i >> 52) & 2047);
    if (e == 0) return 0.0;
    int n = 1076 - bit - e;
    if (n > 0) {
        if (n >= 53) {
            if (i >= 0) i = 0;
            else { int hi = (int)((uint32_t)(2048 + 1024 - bit) << 20); i = (int64_t)((uint64_t)(int64_t)hi << 32); }
        } else {
            int64_t mask = (int64_t)(~0ull << n);
            if (i < 0) i = (int64_t)((uint64_t)i + (uint64_t)~mask);
            i &= mask;
        }
    }
    return f64_from((uint64_t)i);
}

/* dsp_ieee754.h:300-318 / 320-340: add `shift` to the exponent field, unchecked */
static float  shift_float(float f, int shift)  { return f32_from(f32_bits(f) + ((uint32_t)shift << 23)); }
static double shift_double(double d, int shift){ return f64_from(f64_bits(d) + ((uint64_t)(int64_t)shift << 52)); }

/* dsp_fpmath.h:84-98 */
long long oracle_saturate64_031(long long a, int mant)
{
    int64_t pos = (int64_t)(1ull << (mant + 31));
    if (a >= pos) return 0x7FFFFFFFll;
    if (a < -pos) return (long long)0xFFFFFFFF80000000ull;
    return a >> mant;
}

/* dsp_header.h:276-283 with the (double,n,m) signature of dsp_header.c:75-77 */
long long oracle_qnm(double x, int n, int m)
{
    int b = n + m;
    if (m >= b || b > 64 || m < 1) return 0;      /* the macro divides by zero here */
    uint64_t lim = 1ull << (b - m - 1);
    if (x >= (double)lim) return (b >= 64) ? 9223372036854775807ll : (long long)((1ull << (b - 1)) - 1);
    if (-x > (double)lim) return (b >= 64) ? (-9223372036854775807ll - 1) : (long long)(1ull << (b - 1));
    if (b >= 33) return (long long)(x * (double)(1ll << m));
    return (long long)(int)(x * (double)(1l << m));
}

/* ------------------------------------------------------------------------------------------
 * biquad cascade and FIR (the two hot loops)
 * ---------------------------------------------------------------------------------------- */

/* dsp_biquadSTD.h:25-77.  state per section: [acc lo, acc hi, x1, x2, y1, y2]; coef = b0 b1 b2
 * (a1-1) a2 in Q4.28, next section `skip` words further.  Returns the last section's accumulator. */
long long oracle_biquads_int(int xn_in, const int *coef, int *state, int num, int skip)
{
    int64_t xn = xn_in, acc = 0;
    const int64_t sat_hi = 1 << (DSP_MANTBQ - 1);
    short n = (short)num;                          /* callee takes `short num` */
    while (n--) {
        int64_t b0 = coef[0], b1 = coef[1], b2 = coef[2], a1 = coef[3], a2 = coef[4];
        coef += skip;
        acc = ld_i64(state);
        int64_t x1 = state[2], x2 = state[3], y1 = state[4], y2 = state[5];
        acc = wadd(acc, wmul(xn, b0));
        acc = wadd(acc, wmul(x1, b1));
        acc = wadd(acc, wmul(x2, b2));
        acc = wadd(acc, wmul(y1, a1));
        acc = wadd(acc, wmul(y2, a2));
        int32_t hi = (int32_t)(acc >> 32);                         /* :25-32, high word only */
        if (hi >= sat_hi) acc = (sat_hi << 32) - 1;
        else if (hi <= 1 - sat_hi) acc = -(sat_hi << 32);
        st_i64(state, acc);
        state[2] = (int)xn; state[3] = (int)x1; state[5] = (int)y1;
        xn = acc >> DSP_MANTBQ;
        state[4] = (int)xn;
        state += 6;
    }
    return acc;
}

/* dsp_biquadSTD.h:84-119 for a double accumulator (formats 4 and 6) */
double oracle_biquads_double(float xn, const float *coef, float *state, int num, int skip)
{
    double acc = 0.0;
    short n = (short)num;
    while (n--) {
        float b0 = coef[0], b1 = coef[1], b2 = coef[2], a1 = coef[3], a2 = coef[4];
        coef += skip;
        acc = ld_f64((const int *)state);
        float x1 = state[2], x2 = state[3], y1 = state[4], y2 = state[5];
        acc += oracle_mul_float_double(xn, b0);
        acc += oracle_mul_float_double(x1, b1);
        acc += oracle_mul_float_double(x2, b2);
        acc += oracle_mul_float_double(y1, a1);
        acc += oracle_mul_float_double(y2, a2);
        st_f64((int *)state, acc);
        state[2] = xn; state[3] = x1; state[5] = y1;
        float yn = (float)acc;
        state[4] = yn;
        xn = yn;
        state += 6;
    }
    return acc;
}

/* same with a float accumulator in state word 0 only (formats 3 and 5) */
static float biquads_float(float xn, const float *coef, float *state, int num, int skip)
{
    float acc = 0.0f;
    short n = (short)num;
    while (n--) {
        float b0 = coef[0], b1 = coef[1], b2 = coef[2], a1 = coef[3], a2 = coef[4];
        coef += skip;
        acc = state[0];
        float x1 = state[2], x2 = state[3], y1 = state[4], y2 = state[5];
        acc = macc_float_float(acc, xn, b0);
        acc = macc_float_float(acc, x1, b1);
        acc = macc_float_float(acc, x2, b2);
        acc = macc_float_float(acc, y1, a1);
        acc = macc_float_float(acc, y2, a2);
        state[0] = acc;
        state[2] = xn; state[3] = x1; state[5] = y1;
        state[4] = acc;
        xn = acc;
        state += 6;
    }
    return acc;
}

/* dsp_firSTD.h:38-52: y = sum_i coef[i] * x[n-i], the delay line shifted by one on the way */
double oracle_fir_double(float xn, const float *coef, float *state, int num)
{
    double acc = 0.0;
    for (int i = 0; i < num; i++) {
        float prev = state[i];
        state[i] = xn;
        acc += oracle_mul_float_double(xn, coef[i]);
        xn = prev;
    }
    return acc;
}

static float fir_float(float xn, const float *coef, float *state, int num)
{
    float acc = 0.0f;
    for (int i = 0; i < num; i++) {
        float prev = state[i];
        state[i] = xn;
        acc = macc_float_float(acc, xn, coef[i]);
        xn = prev;
    }
    return acc;
}

/* Intended meaning of dsp_firSTD.h:8-35 (the reference code is undefined behaviour and yields
 * compiler-dependent non-FIR results; NOT a parity claim): y = sum coef[i]*x[n-i], 64-bit sum. */
static int64_t fir_int_intended(int xn, const int *coef, int *state, int num)
{
    int64_t acc = 0;
    for (int i = 0; i < num; i++) {
        int prev = state[i];
        state[i] = xn;
        acc = wadd(acc, wmul((int64_t)xn, (int64_t)coef[i]));
        xn = prev;
    }
    return acc;
}

/* ------------------------------------------------------------------------------------------
 * per-program context: what the reference keeps in file-scope globals
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int       dither;      /* dsp_tpdf.h:15-21 */
    int       mask;
    long long mask64;
    int       shift;
} tpdf_t;

struct oracle_ctx {
    int          format;
    dspHeader_t *header;                 /* dspHeaderPtr, dsp_runtime.c:36 */
    int          biquadFreqSkip;         /* :37 */
    int          mantissa;               /* :38 */
    int          samplingFreq, minFreq, maxFreq, numFreq, freqIndex;      /* :103-107 */
    unsigned     delayLineFactor;        /* :108 */
    unsigned     rmsFactorFS;            /* :109 */
    int          biquadFreqOffset;       /* :110 */
    /* dsp_tpdf.h:11-13,23,33 */
    int          tpdfValue, tpdfRandom, tpdfDefaultDither;
    tpdf_t       tpdfGlobal;
    uint32_t     xs[4];
};

static int fmt_alu_int(int f)    { return f == 2; }
static int fmt_alu_64(int f)     { return f == 2 || f == 4 || f == 6; }

oracle_ctx *oracle_new(int format)
{
    if (format < 2 || format > 6) return NULL;
    oracle_ctx *c = (oracle_ctx *)calloc(1, sizeof *c);
    if (c) c->format = format;
    return c;
}
void oracle_free(oracle_ctx *c) { free(c); }
int  oracle_format(const oracle_ctx *c) { return c->format; }

/* dsp_runtime.c:42-59 */
opcode_t *oracle_find_core(opcode_t *code, int numCore)
{
    if (code->op.opcode != DSP_HEADER) return 0;
    opcode_t *p = code;
    int seen = 0;
    for (;;) {
        unsigned skip = p->op.skip;
        if (skip == 0) return seen == 0 ? code : 0;
        if (p->op.opcode == DSP_CORE && ++seen == numCore) return p;
        p += skip;
    }
}

/* dsp_runtime.c:62-77 */
opcode_t *oracle_find_core_begin(opcode_t *p)
{
    if (p && p->op.opcode == DSP_CORE)
        for (;;) {
            unsigned skip = p->op.skip;
            int op = p->op.opcode;
            if (skip == 0) return p;
            if (op == DSP_CORE || op == DSP_NOP || op == DSP_PARAM || op == DSP_PARAM_NUM) p += skip;
            else break;
        }
    return p;
}

/* ---- TPDF / PRNG: dsp_tpdf.h ---- */
static inline uint32_t rotl32(uint32_t x, unsigned k) { return (x << k) | (x >> (32 - k)); }

static uint32_t xoshiro128p(uint32_t *s)          /* dsp_tpdf.h:35-49 */
{
    uint32_t r = s[0] + s[3], t = s[1] << 9;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl32(s[3], 11);
    return r;
}

/* dsp_tpdf.h:55-80: returns 1 when nothing had to change */
static int tpdf_prepare(oracle_ctx *c, const tpdf_t *current, tpdf_t *local, int dith)
{
    if (dith == 0) dith = c->tpdfDefaultDither;
    if (dith == current->dither) return 1;
    local->dither = dith;
    local->mask = (int)(0xFFFFFFFFu << ((32 - dith) & 31));
    if (fmt_alu_int(c->format)) {
        local->mask64 = (long long)wshl((int64_t)local->mask, DSP_MANT);
        local->shift = DSP_MANT - dith + 1;
    }
    return 0;
}

static void tpdf_init(oracle_ctx *c, int seed, int defaultDither)   /* dsp_tpdf.h:85-99 */
{
    c->tpdfRandom = seed;
    c->tpdfValue = 0;
    c->tpdfDefaultDither = defaultDither;
    c->tpdfGlobal.dither = -1;
    tpdf_prepare(c, &c->tpdfGlobal, &c->tpdfGlobal, 0);
    uint32_t s = (uint32_t)seed;
    c->xs[0] = s | 1;
    c->xs[1] = rotl32(s | 8, 7);
    c->xs[2] = rotl32(s | 16, 11);
    c->xs[3] = rotl32(s | 24, 17);
}

static int tpdf_calc(oracle_ctx *c)                /* dsp_tpdf.h:103-130 */
{
    int r1 = (int)xoshiro128p(c->xs);
    int r2 = (int)xoshiro128p(c->xs);
    c->tpdfRandom = r2;
    int r = (r1 >> 1) + (r2 >> 1);
    c->tpdfValue = r;
    return r;
}

/* ---- dspRuntimeReset: dsp_runtime.c:116-145 ---- */
static const double k_delay_factor = 4294.967296;    /* 2^32 / 10^6, :81 */

int oracle_reset(oracle_ctx *c, int fs, int random, int defaultDither)
{
    int idx = dspConvertFrequencyToIndex(fs);
    if (idx >= FMAXpos) return -1;
    int mn = c->header->freqMin, mx = c->header->freqMax;
    if (idx < mn || idx > mx) return -2;
    c->samplingFreq = idx; c->minFreq = mn; c->maxFreq = mx;
    c->freqIndex = idx - mn;
    c->numFreq = mx - mn + 1;
    c->biquadFreqSkip = 2 + 6 * c->numFreq;
    c->biquadFreqOffset = 5 + 6 * c->freqIndex;
    c->delayLineFactor = (unsigned)(k_delay_factor * (double)dspConvertFrequencyFromIndex(idx));  /* :82-90 */
    c->rmsFactorFS = (unsigned)(1000.0 / (double)dspConvertFrequencyFromIndex(idx));             /* :92-101 */
    int *data = (int *)c->header + c->header->totalLength;
    for (int i = 0; i < c->header->dataSize; i++) data[i] = 0;
    tpdf_init(c, random, defaultDither);
    return 0;
}

/* ---- dspChangeFormat: dsp_runtime.c:198-299 (with the compile fix of :208 that passes
 * (x, 32-new, new) to dspQNM).  numFreq is whatever the previous Reset left (0 on first use),
 * exactly as in the reference, so BIQUADS coefficients are NOT converted on a fresh load.   */
static void change_datum(opcode_t *w, int oldf, int newf)
{
    if (oldf) {
        if (newf) {
            int d = newf - oldf;
            if (d > 0) w->i32 = (int)((uint32_t)w->i32 << d);
            if (d < 0) w->i32 >>= -d;
        } else w->f32 = (float)w->i32 / (float)(1 << oldf);
    } else if (newf) w->i32 = (int)oracle_qnm(w->f32, 32 - newf, newf);
}

static void change_format(oracle_ctx *c, opcode_t *prog, int newf)
{
    dspHeader_t *h = (dspHeader_t *)prog;
    int oldf = h->format;
    if (oldf == newf) return;
    opcode_t *p = prog;
    for (;;) {
        unsigned skip = p->op.skip;
        if (skip == 0) break;
        opcode_t *a = p + 1;
        switch (p->op.opcode) {
        case DSP_DIRAC: case DSP_SQUAREWAVE:
            a++;                                   /* fallthrough: immediate after a data pointer */
        case DSP_MUL_VALUE: case DSP_DIV_VALUE: case DSP_DATA_TABLE: case DSP_CLIP:
            change_datum(a, oldf, newf); break;
        case DSP_LOAD_GAIN:
            a++;                                   /* fallthrough: skip IO */
        case DSP_GAIN: case DSP_SAT0DB_GAIN: case DSP_SAT0DB_TPDF_GAIN:
            change_datum(p + a->i32, oldf, newf); break;
        case DSP_LOAD_MUX: {
            opcode_t *t = p + a->i32;
            short n = (short)t->i32;
            t++;
            for (int i = 0; i < n; i++) { t++; change_datum(t++, oldf, newf); }
            break; }
        case DSP_BIQUADS: {
            a++;
            opcode_t *t = p + a->i32;
            short ns = (short)t->i32;
            t += 3;
            for (int i = 0; i < ns; i++) {
                t += 2;
                for (int j = 0; j < c->numFreq; j++) { for (int k = 0; k < 5; k++) change_datum(t++, oldf, newf); t++; }
            }
            break; }
        case DSP_DITHER_NS2: {
            a++;
            opcode_t *t = p + a->i32;
            for (int i = 0; i < 3 * c->numFreq; i++) change_datum(t++, oldf, newf);
            break; }
        case DSP_DCBLOCK:
            a++;
            for (int i = 0; i < c->numFreq; i++) change_datum(a++, oldf, newf);
            break;
        case DSP_SINE:
            a++;
            change_datum(a++, oldf, newf);
            for (int i = 0; i < c->numFreq; i++) change_datum(a++, oldf, newf);
            break;
        default: break;                            /* DSP_FIR: "TODO" in the reference, :267 */
        }
        p += skip;
    }
    h->format = (unsigned short)newf;
}

/* ---- dspRuntimeInit: dsp_runtime.c:150-195 ---- */
int oracle_init(oracle_ctx *c, opcode_t *code, int maxSize, int fs, int random, int defaultDither)
{
    c->header = (dspHeader_t *)code;
    if (code->op.opcode != DSP_HEADER) return -1;
    int length = c->header->totalLength, size = c->header->dataSize;
    if (size + length > maxSize) return -6;
    unsigned sum; int cores;
    avdspChecksumWalk(code, (unsigned)length, &sum, &cores);
    if (cores < 1) return -3;
    if (sum != c->header->checkSum) return -4;
    if (c->header->maxOpcode >= DSP_MAX_OPCODE) return -5;
    c->mantissa = DSP_MANT;
    if (fmt_alu_int(c->format)) { if (c->header->format != DSP_MANT) change_format(c, code, DSP_MANT); }
    else                        { if (c->header->format != 0)        change_format(c, code, 0); }
    if (fs) { int r = oracle_reset(c, fs, random, defaultDither); if (r) return r; }
    return length;
}

/* ------------------------------------------------------------------------------------------
 * the interpreter, instantiated once per arithmetic model
 * ---------------------------------------------------------------------------------------- */
#define ORC_FMT 2
#define ORC_STATE_INDEX(v, alloc) ((unsigned)(v) < (unsigned)(alloc) ? (v) : 0)
#include "oracle_interp.inc"
#undef ORC_FMT
#define ORC_FMT 3
#include "oracle_interp.inc"
#undef ORC_FMT
#define ORC_FMT 4
#include "oracle_interp.inc"
#undef ORC_FMT
#define ORC_FMT 5
#include "oracle_interp.inc"
#undef ORC_FMT
#define ORC_FMT 6
#include "oracle_interp.inc"
#undef ORC_FMT

/* The reference is built -Ofast (CMakeLists.txt:7, runtime/Makefile:13): its shared object carries gcc's
 * crtfastmath start-up code, which sets MXCSR.FTZ and MXCSR.DAZ in the thread that loads it.  Every SSE
 * arithmetic or conversion instruction of the runtime therefore reads subnormal operands as signed zero and
 * flushes subnormal results to signed zero -- the tail of every decaying filter goes through that.  The
 * oracle is compiled without fast-math and runs its frames under the same two MXCSR bits instead. */
#if defined(__x86_64__) || defined(__i386__)
  #include <xmmintrin.h>
  #define ORC_FTZ_DAZ_ON(saved)  do { (saved) = _mm_getcsr(); _mm_setcsr((saved) | 0x8040u); } while (0)
  #define ORC_FTZ_DAZ_OFF(saved) _mm_setcsr(saved)
#else
  #error "the oracle mirrors the x86 MXCSR flush modes of the reference build; port ORC_FTZ_DAZ_* for this host"
#endif

int oracle_run(oracle_ctx *c, opcode_t *core, int *rundata, void *samples)
{
    unsigned saved;
    int rc = -1;
    ORC_FTZ_DAZ_ON(saved);
    switch (c->format) {
    case 2: rc = run_frame_2(c, core, rundata, (int *)samples); break;
    case 3: rc = run_frame_3(c, core, rundata, (int *)samples); break;
    case 4: rc = run_frame_4(c, core, rundata, (int *)samples); break;
    case 5: rc = run_frame_5(c, core, rundata, (float *)samples); break;
    case 6: rc = run_frame_6(c, core, rundata, (float *)samples); break;
    }
    ORC_FTZ_DAZ_OFF(saved);
    return rc;
}

/* the host loop over one block with the caller's samples[] array: what the frame holds outside the two
 * windows stays from frame to frame and from call to call (linux/avdsp_plugin.c:93 inputOutput[]) */
int oracle_run_block_frame(oracle_ctx *c, opcode_t *core, int *rundata,
                           const void *in, int in_stride, int in_io_base,
                           void *out, int out_stride, int out_io_base,
                           int nframes, void *frame)
{
    /* samples are 32 bits wide in every format, so the gather/scatter is format-agnostic */
    uint32_t *scratch = (uint32_t *)frame;
    const uint32_t *src = (const uint32_t *)in;
    uint32_t *dst = (uint32_t *)out;
    for (int n = 0; n < nframes; n++) {
        memcpy(scratch + out_io_base, dst + (size_t)n * out_stride, (size_t)out_stride * 4);
        memcpy(scratch + in_io_base,  src + (size_t)n * in_stride,  (size_t)in_stride * 4);
        oracle_run(c, core, rundata, scratch);
        memcpy(dst + (size_t)n * out_stride, scratch + out_io_base, (size_t)out_stride * 4);
    }
    return 0;
}

int oracle_run_block(oracle_ctx *c, opcode_t *core, int *rundata,
                     const void *in, int in_stride, int in_io_base,
                     void *out, int out_stride, int out_io_base,
                     int nframes, int scratch_len)
{
    uint32_t *scratch = (uint32_t *)calloc((size_t)scratch_len, 4);
    if (!scratch) return -1;
    int rc = oracle_run_block_frame(c, core, rundata, in, in_stride, in_io_base, out, out_stride, out_io_base, nframes, scratch);
    free(scratch);
    return rc;
}
