/*
 * avdsp_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the AVDSP runtime (module_avdsp/runtime/dsp_runtime.c and the headers it
 * includes) used ONLY as the checker for the HIP path: tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load liboracle.so; nothing under avdsp_amd/ does.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_golden.py replays the golden vectors that
 * tests/golden/make_goldens.py produced by running the compiled reference (oracle/build_ref.sh ->
 * oracle/_ref/) on seeded inputs, and tests/test_oracle_vs_ref.py compares against oracle/_ref/
 * directly whenever it is present.
 *
 * Differences from the reference, all deliberate:
 *   - no process-wide globals: every loaded program is an oracle_ctx, so several can live in one
 *     process (needed for the multi-threaded CPU baseline); the reference keeps one program per
 *     process (dsp_runtime.c:36-38,103-110, dsp_tpdf.h:11-13,23,33).
 *   - the arithmetic model (DSP_FORMAT 2..6) is a run-time field, not a compile-time macro.
 *   - behaviour that is undefined in the reference (int-mode FIR dsp_firSTD.h:8-35, shifts by
 *     >= word size in dsps31Double0DB for |x| < 2^-42, DIRAC/SQUAREWAVE/SINE in int mode which call
 *     functions that do not exist) is given the evident intended meaning and is not claimed as parity.
 */
#ifndef AVDSP_ORACLE_H_
#define AVDSP_ORACLE_H_

#include "../include/avdsp_format.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_ctx oracle_ctx;

/* format = 2,3,4,5,6 (DSP_FORMAT_*); returns NULL for anything else */
oracle_ctx *oracle_new(int format);
void        oracle_free(oracle_ctx *ctx);
int         oracle_format(const oracle_ctx *ctx);

/* dspRuntimeInit / dspRuntimeReset (dsp_runtime.c:116-195): same arguments, same return codes */
int oracle_init(oracle_ctx *ctx, opcode_t *code, int maxSize, int fs, int random, int defaultDither);
int oracle_reset(oracle_ctx *ctx, int fs, int random, int defaultDither);

/* dspFindCore / dspFindCoreBegin (dsp_runtime.c:42-77) */
opcode_t *oracle_find_core(opcode_t *code, int numCore);
opcode_t *oracle_find_core_begin(opcode_t *ptr);

/* dspRuntime_N (dsp_runtime.c:302-1314): one frame; samples is int32[] or float[] per the format */
int oracle_run(oracle_ctx *ctx, opcode_t *core, int *rundata, void *samples);

/*
 * nframes successive oracle_run() calls over frame-interleaved buffers, i.e. the host loop of
 * linux/avdsp_plugin.c:98-141 for 32-bit samples: for frame n the scratch frame holds
 * samples[in_io_base + k] = in[n*in_stride + k] and afterwards out[n*out_stride + k] =
 * samples[out_io_base + k].  Output slots the core never stores keep their previous content.
 * scratch_len = size of the scratch frame (must exceed every IO number the core touches).
 */
/* same loop over a frame the caller keeps (slots outside the windows persist between frames and calls) */
int oracle_run_block_frame(oracle_ctx *ctx, opcode_t *core, int *rundata,
                           const void *in, int in_stride, int in_io_base,
                           void *out, int out_stride, int out_io_base,
                           int nframes, void *frame);
int oracle_run_block(oracle_ctx *ctx, opcode_t *core, int *rundata,
                     const void *in, int in_stride, int in_io_base,
                     void *out, int out_stride, int out_io_base,
                     int nframes, int scratch_len);

/* exposed pieces of the arithmetic, for unit tests against the reference's own headers */
double   oracle_mul_float_double(float a, float b);          /* dsp_ieee754.h:377-410 */
float    oracle_mul_float_float(float a, float b);           /* dsp_ieee754.h:342-375 */
float    oracle_int_to_float_scaled(int x, int shift);       /* dsp_ieee754.h:204-250 */
double   oracle_int_to_double_scaled(int x, int shift);      /* dsp_ieee754.h:253-298 */
int      oracle_s31_from_double(double d);                   /* dsp_ieee754.h:85-107  */
int      oracle_s31_from_float(float f);                     /* dsp_ieee754.h:60-83   */
double   oracle_saturate_double(double d);                   /* dsp_ieee754.h:187-199 */
float    oracle_saturate_float(float f);                     /* dsp_ieee754.h:171-184 */
double   oracle_truncate_double(double d, int bit);          /* dsp_ieee754.h:142-168 */
float    oracle_truncate_float(float f, int bit);            /* dsp_ieee754.h:112-139 */
long long oracle_saturate64_031(long long a, int mant);      /* dsp_fpmath.h:84-98    */
long long oracle_biquads_int(int xn, const int *coef, int *state, int num, int skip);      /* dsp_biquadSTD.h:34-77  */
double   oracle_biquads_double(float xn, const float *coef, float *state, int num, int skip); /* dsp_biquadSTD.h:84-119 */
double   oracle_fir_double(float xn, const float *coef, float *state, int num);            /* dsp_firSTD.h:38-52 */
long long oracle_qnm(double x, int n, int m);                /* dsp_header.h:276-283, dsp_header.c:75-77 */

#ifdef __cplusplus
}
#endif
#endif
