/*
 * avdsp_encoder.h -- C API of the program encoder (host-only, no GPU involved): the step in front of
 * the hot path, SURVEY.md 8(f) rank 1.  Same names, argument meaning and error behaviour as the
 * reference's encoder (module_avdsp/encoder/dsp_encoder.h:17-223, dsp_filters.h:14-76,
 * dsp_fileaccess.h:38): a program written for the reference -- a sequence of dsp_XXX() calls between
 * dspEncoderInit() and dsp_END_OF_CODE() -- compiles against this header (or the forwarding headers in
 * include/compat/) and produces the same words, byte for byte (tests/test_encoder.py: the reference's
 * reproducible osx .bin programs and the reference-encoded opcode tour).
 *
 * Deliberate differences:
 *   - dsp_FIR() points each rate at its impulse's LENGTH word; the reference points one word early
 *     (encoder/dsp_encoder.c:1311-1313 reads the section header as the first impulse), which its own
 *     runtime then misreads.  dspFir_ImpulseData() adds taps from memory.
 *   - the nanoSHARC import and the text dump files are not provided; dsp_dumpParameter*() accept and
 *     ignore (they never change the program words).  (dsp_Hilbert() is provided since round 5:
 *     dsp_filters.h:76, the design of encoder/dsp_HilbertDesign.c restated in avdsp_encoder.c.)
 * Errors: like the reference, a malformed program prints "FATAL ERROR : ..." and exit(1)s.
 */
#ifndef AVDSP_ENCODER_H_
#define AVDSP_ENCODER_H_

#include <math.h>
#include <stdio.h>
#include "avdsp_format.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef float  dspGainParam_t;          /* dsp_header.h:192 */
typedef double dspFilterParam_t;        /* dsp_header.h:193 */

/* programs print through these (dsp_header.h:17-33); quiet unless DSP_PRINTF is defined */
#if defined(DSP_PRINTF) && DSP_PRINTF
  #define dspprintf(...)  { printf(__VA_ARGS__); }
#else
  #define dspprintf(...)  { }
#endif
#define dspprintf1(...) dspprintf(__VA_ARGS__)
#define dspprintf2(...) dspprintf(__VA_ARGS__)
#define dspprintf3(...) dspprintf(__VA_ARGS__)

/* dsp_header.h:270-285 as functions (dsp_header.c:75-85) */
long long dspQNM(double x, int n, int m);
long long dspQM64(double x, int m);
int       dspQM32(double x, int m);
#define DSP_QNM(x, n, m) dspQNM((x), (n), (m))
#define DSP_QM32(x, m)   dspQM32((x), (m))
#define DSP_QM64(x, m)   dspQM64((x), (m))

extern dspHeader_t *dspHeaderPtr;
extern int dspMinSamplingFreq, dspMaxSamplingFreq;

/* ---- dsp_encoder.h:17-31 ---- */
void dspEncoderFormat(int format);
void dspEncoderInit(opcode_t *opcodeTable, int max, int format, int minFreq, int maxFreq, int maxIO);
void dsp_dumpParameter(int addr, int size, char *name);
void dsp_dumpParameterNum(int addr, int size, char *name, int num);
void setSerialHash(unsigned hash);
int  addCode(int code);
int  addFloat(float value);
int  opcodeIndex(void);
int  opcodeIndexAligned8(void);
int  opcodeIndexMisAligned8(void);

/* ---- dsp_encoder.h:33-223 ---- */
int  dsp_END_OF_CODE(void);
void dsp_NOP(void);
void dsp_CORE(void);
void dsp_SERIAL(unsigned hash);
void dsp_SWAPXY(void);
void dsp_COPYXY(void);
void dsp_COPYYX(void);
void dsp_CLRXY(void);
void dsp_ADDXY(void);
void dsp_ADDYX(void);
void dsp_SUBXY(void);
void dsp_SUBYX(void);
void dsp_MULXY(void);
void dsp_DIVXY(void);
void dsp_DIVYX(void);
void dsp_AVGXY(void);
void dsp_AVGYX(void);
void dsp_SQRTX(void);
void dsp_NEGX(void);
void dsp_NEGY(void);
int  dsp_TPDF_CALC(int bits);
int  dsp_TPDF(int bits);
void dsp_WHITE(void);
void dsp_SAT0DB(void);
void dsp_SAT0DB_GAIN(int paramAddr);
void dsp_SAT0DB_GAIN_Fixed(dspGainParam_t gain);
void dsp_SAT0DB_TPDF(void);
void dsp_SAT0DB_TPDF_GAIN(int paramAddr);
void dsp_SAT0DB_TPDF_GAIN_Fixed(dspGainParam_t gain);
void dsp_SHIFT(int bits);
void dsp_SHIFT_FixedInt(int bits);
void dsp_LOAD(int IO);
void dsp_LOAD_GAIN(int IO, int paramAddr);
void dsp_LOAD_GAIN_Fixed(int IO, dspGainParam_t gain);
int  dsp_LOAD_MUX(int paramAddr);
int  dspLoadMux_Inputs(int number);
void dspLoadMux_Data(int in, dspGainParam_t gain);
void dsp_STORE(int IO);
void dsp_LOAD_STORE(void);
void dspLoadStore_Data(int memin, int memout);
void dsp_LOAD_MEM(int paramAddr);
void dsp_STORE_MEM(int paramAddr);
void dsp_LOAD_MEM_Index(int paramAddr, int index);
void dsp_STORE_MEM_Index(int paramAddr, int index);
int  dspMem_Location(void);
int  dspMem_LocationMultiple(int number);
int  dsp_PARAM(void);
int  dsp_PARAM_NUM(int num);
int  dspDataTableInt(int *data, int n);
int  dspDataTableFloat(float *data, int n);
int  dspData2(int a, int b);
int  dspData4(int a, int b, int c, int d);
int  dspData6(int a, int b, int c, int d, int e, int f);
int  dspData8(int a, int b, int c, int d, int e, int f, int g, int h);
int  dspGenerator_Sine(int samples);
void dsp_GAIN_Fixed(dspGainParam_t gain);
void dsp_GAIN(int paramAddr);
int  dspGain_Default(dspGainParam_t gain);
void dsp_VALUE_Fixed(float value);
void dsp_VALUE_FixedInt(int value);
void dsp_VALUE(int paramAddr);
int  dspValue_Default(float value);
void dsp_DIV_Fixed(float value);
void dsp_DIV_FixedInt(int value);
void dsp_MUL_Fixed(float value);
void dsp_MUL_FixedInt(int value);
void dsp_AND_FixedInt(int value);
void dsp_DELAY(int paramAddr);
void dsp_DELAY_1(void);
int  dspDelay_MicroSec_Max(int maxus);
int  dspDelay_MicroSec_Max_Default(int maxus, int us);
int  dspDelay_MilliMeter_Max(int maxmm, float speed);
int  dspDelay_MilliMeter_Max_Default(int maxmm, int mm, float speed);
void dsp_DELAY_FixedMicroSec(int microSec);
void dsp_DELAY_FixedMilliMeter(int mm, float speed);
void dsp_DELAY_DP(int paramAddr);
void dsp_DELAY_DP_FixedMicroSec(int microSec);
void dsp_DELAY_DP_FixedMilliMeter(int mm, float speed);
void dsp_DATA_TABLE(int paramAddr, dspGainParam_t gain, int divider, int size);
int  dsp_BIQUADS(int paramAddr);
int  dspBiquad_Sections(int number);
int  dspBiquad_Sections_Flexible(void);
int  dspBiquad_Sections_Maximum(int number);
void dsp_FIR(int paramAddr);
int  dspFir_Impulses(void);
int  dspFir_Delay(int value);
int  dspFir_ImpulseFile(char *name, int length);
int  dspFir_ImpulseData(const float *taps, int length);          /* extension: taps from memory */
void dsp_RMS(int timems, int delayLine);
void dsp_RMS_MilliSec(int timems, int delayms);
void dsp_PWRXY(int timems, int delayLine);
void dsp_PWRXY_MilliSec(int timems, int delayms);
void dsp_DCBLOCK(int lowfreq);
void dsp_DITHER(void);
void dsp_DITHER_NS2(int paramAddr);
void dsp_DISTRIB(int IO, int size);
void dsp_DIRAC_Fixed(int freq, dspGainParam_t gain);
void dsp_SQUAREWAVE_Fixed(int freq, dspGainParam_t gain);
void dsp_CLIP_Fixed(dspGainParam_t value);
void dsp_SINE_Fixed(int freq, dspGainParam_t gain);

static inline dspGainParam_t dB2gain(dspGainParam_t db) { db /= 20.0; return pow(10, db); }   /* dsp_encoder.h:219-221 */

/* ---- dsp_filters.h:14-76 ---- */
enum filterTypes {
    BEna1, LPBE2, LPBE3, LPBE4, LPBE5, LPBE6, LPBE7, LPBE8,
    BEna2, HPBE2, HPBE3, HPBE4, HPBE5, HPBE6, HPBE7, HPBE8,
    BEna3, LPBE3db2, LPBE3db3, LPBE3db4, LPBE3db5, LPBE3db6, LPBE3db7, LPBE3db8,
    BEna4, HPBE3db2, HPBE3db3, HPBE3db4, HPBE3db5, HPBE3db6, HPBE3db7, HPBE3db8,
    BUna1, LPBU2, LPBU3, LPBU4, LPBU5, LPBU6, LPBU7, LPBU8,
    BUna2, HPBU2, HPBU3, HPBU4, HPBU5, HPBU6, HPBU7, HPBU8,
    Fna1, LPLR2, LPLR3, LPLR4, Fna3, LPLR6, Fna4, LPLR8,
    Fna5, HPLR2, HPLR3, HPLR4, Fna7, HPLR6, Fna8, HPLR8,
    FLP1, FLP2, FHP1, FHP2, FLS1, FLS2, FHS1, FHS2,
    FAP1, FAP2, FPEAK, FNOTCH, FBP0DB, FBPQ, FHILB
};
int dsp_Filter2ndOrder(int type, dspFilterParam_t freq, dspFilterParam_t Q, dspGainParam_t gain);
int dsp_Filter1stOrder(int type, dspFilterParam_t freq, dspGainParam_t gain);
int dsp_LP_BES2(dspFilterParam_t freq);      int dsp_HP_BES2(dspFilterParam_t freq);
int dsp_LP_BES2_3DB(dspFilterParam_t freq);  int dsp_HP_BES2_3DB(dspFilterParam_t freq);
int dsp_LP_BUT2(dspFilterParam_t freq);      int dsp_HP_BUT2(dspFilterParam_t freq);
int dsp_LP_LR2(dspFilterParam_t freq);       int dsp_HP_LR2(dspFilterParam_t freq);
int dsp_LP_BES3(dspFilterParam_t freq);      int dsp_HP_BES3(dspFilterParam_t freq);
int dsp_LP_BES3_3DB(dspFilterParam_t freq);  int dsp_HP_BES3_3DB(dspFilterParam_t freq);
int dsp_LP_BUT3(dspFilterParam_t freq);      int dsp_HP_BUT3(dspFilterParam_t freq);
int dsp_LP_LR3(dspFilterParam_t freq);       int dsp_HP_LR3(dspFilterParam_t freq);
int dsp_LP_BES4(dspFilterParam_t freq);      int dsp_HP_BES4(dspFilterParam_t freq);
int dsp_LP_BES4_3DB(dspFilterParam_t freq);  int dsp_HP_BES4_3DB(dspFilterParam_t freq);
int dsp_LP_BUT4(dspFilterParam_t freq);      int dsp_HP_BUT4(dspFilterParam_t freq);
int dsp_LP_LR4(dspFilterParam_t freq);       int dsp_HP_LR4(dspFilterParam_t freq);
int dsp_LP_BES6(dspFilterParam_t freq);      int dsp_HP_BES6(dspFilterParam_t freq);
int dsp_LP_BES6_3DB(dspFilterParam_t freq);  int dsp_HP_BES6_3DB(dspFilterParam_t freq);
int dsp_LP_BUT6(dspFilterParam_t freq);      int dsp_HP_BUT6(dspFilterParam_t freq);
int dsp_LP_LR6(dspFilterParam_t freq);       int dsp_HP_LR6(dspFilterParam_t freq);
int dsp_LP_BES8(dspFilterParam_t freq);      int dsp_HP_BES8(dspFilterParam_t freq);
int dsp_HP_BES8_3DB(dspFilterParam_t freq);
int dsp_LP_BUT8(dspFilterParam_t freq);      int dsp_HP_BUT8(dspFilterParam_t freq);
int dsp_LP_LR8(dspFilterParam_t freq);       int dsp_HP_LR8(dspFilterParam_t freq);
int dsp_filter(int type, dspFilterParam_t freq, dspFilterParam_t Q, dspGainParam_t gain);
/* dsp_filters.h:76: one branch of a 90-degree phase splitter as a bank of `stages` (1 .. 10) all-pass cells (c - z^-2)/(1 - c z^-2);
 * transition = width of the transition band in Hz; phase 0: the reference branch, else the +90 degree branch */
int dsp_Hilbert(int stages, dspFilterParam_t transition, dspGainParam_t phase);

/* ---- dsp_fileaccess.h:38,41: program words to / from a binary file; returns the word count or -1 ---- */
int dspCreateBuffer(char *name, int *buff, int size);
int dspReadBuffer(char *name, int *buff, int size);

#ifdef __cplusplus
}
#endif
#endif /* AVDSP_ENCODER_H_ */
