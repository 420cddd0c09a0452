/*
 * ref_kernels.c -- TEST INFRASTRUCTURE.  Thin exports around the reference's own hot kernels,
 * compiled by oracle/build_ref.sh with -I/root/reference/module_avdsp/runtime and
 * -DDSP_FORMAT={2,6}.  The headers are included UNMODIFIED from where they lie; this file adds no
 * arithmetic of its own.  Output: oracle/_ref/refk_N.so, used by tests/test_oracle_vs_ref.py and
 * tests/golden/make_goldens.py to pin oracle/avdsp_oracle.c kernel by kernel.
 */
#define DSP_IEEE754_OPTIMISE 63          /* as dsp_runtime.c:10 */
#include "dsp_runtime.h"
#include "dsp_ieee754.h"
#include "dsp_fpmath.h"
#include "dsp_biquadSTD.h"
#include "dsp_firSTD.h"

int refk_format(void) { return DSP_FORMAT; }

#if DSP_FORMAT == 2
long long refk_biquads_int(int xn, int *coef, int *state, int num, int skip)
{ return dsp_calc_biquads_int(xn, coef, state, (short)num, DSP_MANTBQ, skip); }
long long refk_saturate64_031(long long a, int mant) { dspSaturate64_031(&a, mant); return a; }
int       refk_shift_int(long long a, int mant)      { return dspShiftInt(a, mant); }
long long refk_macs_0(int a, int b)                  { long long r; dspmacs64_32_32_0(&r, a, b); return r; }
unsigned  refk_mulu32(unsigned a, unsigned b)        { return dspmulu32_32_32(a, b); }
#else
double refk_biquads_double(float xn, float *coef, float *state, int num, int skip)
{ return dsp_calc_biquads_float(xn, coef, state, (short)num, skip); }
double refk_fir_double(float xn, float *coef, float *state, int num)
{ return dsp_calc_fir_float(xn, coef, state, num); }
double refk_mul_float_double(float a, float b)       { return dspMulFloatDouble(a, b); }
float  refk_mul_float_float(float a, float b)        { return dspMulFloatFloat(a, b); }
float  refk_int_to_float_scaled(int x, int shift)    { return dspIntToFloatScaled(x, shift); }
double refk_int_to_double_scaled(int x, int shift)   { return dspIntToDoubleScaled(x, shift); }
int    refk_s31_from_double(double d)                { return dsps31Double0DB(d); }
int    refk_s31_from_float(float f)                  { return dsps31Float0DB(f); }
double refk_saturate_double(double d)                { dspSaturateDouble0db(&d); return d; }
float  refk_saturate_float(float f)                  { dspSaturateFloat0db(&f); return f; }
double refk_truncate_double(double d, int bit)       { dspTruncateDouble0DB(&d, bit); return d; }
float  refk_truncate_float(float f, int bit)         { dspTruncateFloat0DB(&f, bit); return f; }
#endif
