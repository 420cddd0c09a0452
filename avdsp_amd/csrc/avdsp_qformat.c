/*
 * avdsp_qformat.c -- fixed-point parameter conversion shared by the runtime library and the encoder
 * library (the reference compiles runtime/dsp_header.c into both as well).
 */
#include "avdsp_format.h"

long long dspQNM(double x, int n, int m);
long long dspQM64(double x, int m);
int       dspQM32(double x, int m);

/* dsp_header.h:276-285 + dsp_header.c:75-85: saturating double -> Qn.m */
long long dspQNM(double x, int n, int m)
{
    int b = n + m;
    if (m >= b || b > 64 || m < 1) return 0;
    double lim = (double)(1ull << (b - m - 1));
    if (x >= lim) return b >= 64 ? 9223372036854775807ll : (long long)((1ull << (b - 1)) - 1);
    if (-x > lim) return b >= 64 ? (-9223372036854775807ll - 1) : (long long)(1ull << (b - 1));
    if (b >= 33) return (long long)(x * (double)(1ll << m));
    return (long long)(int)(x * (double)(1l << m));
}
long long dspQM64(double x, int m) { return dspQNM(x, 64 - m, m); }
int       dspQM32(double x, int m) { return (int)dspQNM(x, 32 - m, m); }
