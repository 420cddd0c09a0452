/* compat/dsp_runtime.h -- source compatibility for hosts written against the reference's runtime header
 * (module_avdsp/runtime/dsp_runtime.h:24-131,160-164).  A reference host is compiled with -DDSP_FORMAT=N and calls
 * DSP_RUNTIME_FORMAT(dspRuntime)(core, rundata, samples) with dspSample_t samples; this header gives it the same
 * names on top of include/avdsp_runtime.h, whose library carries the _2 .. _6 entry points side by side.
 *
 *   DSP_FORMAT   samples (dspSample_t)   accumulator (dspALU_t)   state (dspALU_SP_t)   parameters (dspParam_t)
 *       2            int                    long long                int                   int  (Q4.28)
 *       3            int                    float                    float                 float
 *       4            int                    double                   float                 float
 *       5            float                  float                    float                 float
 *       6            float                  double                   float                 float
 */
#ifndef AVDSP_COMPAT_DSP_RUNTIME_H_
#define AVDSP_COMPAT_DSP_RUNTIME_H_

#include "../avdsp_runtime.h"

#ifndef DSP_FORMAT
#define DSP_FORMAT 2                     /* the reference's default (dsp_runtime.h:24-26) */
#endif

#if DSP_FORMAT < 2 || DSP_FORMAT > 6
#error "DSP_FORMAT must be 2 .. 6 (1, the 32-bit integer model, is an #error in the reference as well)"
#endif

#define AVDSP_PASTE2_(a, b) a##_##b
#define AVDSP_PASTE_(a, b)  AVDSP_PASTE2_(a, b)
#define DSP_RUNTIME_FORMAT(name) AVDSP_PASTE_(name, DSP_FORMAT)

#define DSP_ALU_INT      (DSP_FORMAT == 2)
#define DSP_ALU_FLOAT    (DSP_FORMAT != 2)
#define DSP_ALU_64B      (DSP_FORMAT == 2 || DSP_FORMAT == 4 || DSP_FORMAT == 6)
#define DSP_SAMPLE_INT   (DSP_FORMAT <= 4)
#define DSP_SAMPLE_FLOAT (DSP_FORMAT >= 5)

#if DSP_SAMPLE_FLOAT
typedef float dspSample_t;
#else
typedef int dspSample_t;
#endif

#if DSP_FORMAT == 2
typedef long long dspALU_t;
typedef int       dspALU_SP_t;
typedef int       dspParam_t;
#elif DSP_FORMAT == 3 || DSP_FORMAT == 5
typedef float dspALU_t;
typedef float dspALU_SP_t;
typedef float dspParam_t;
#else
typedef double dspALU_t;
typedef float  dspALU_SP_t;
typedef float  dspParam_t;
#endif

#endif /* AVDSP_COMPAT_DSP_RUNTIME_H_ */
