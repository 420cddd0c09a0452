/*
 * avdsp_format.h -- wire format of an AVDSP encoded program (.bin), as consumed by
 * dspRuntimeInit()/dspRuntime_N().
 *
 * This is a from-scratch statement of the binary contract defined by the reference in
 *   module_avdsp/runtime/dsp_header.h:40-132   (opcode numbering)
 *   module_avdsp/runtime/dsp_header.h:136-187  (sample-rate index table)
 *   module_avdsp/runtime/dsp_header.h:197-228  (opcode word, 12-word header)
 *   module_avdsp/runtime/dsp_header.h:234-251  (checksum walk)
 * All words are 32-bit little endian.  An opcode head word is (opcode << 16) | skip, where
 * skip = distance in words to the next head word and skip == 0 terminates the program.
 *
 * The types opcode_t / dspHeader_t keep the reference's names and layout because they appear
 * in the C API that hosts bind to (see avdsp_runtime.h).
 */
#ifndef AVDSP_FORMAT_H_
#define AVDSP_FORMAT_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- opcode numbering: dsp_header.h:40-132 (enum order is the contract) ---- */
enum {
    DSP_END_OF_CODE = 0, DSP_HEADER = 1, DSP_NOP = 2, DSP_CORE = 3, DSP_PARAM = 4,
    DSP_PARAM_NUM = 5, DSP_SERIAL = 6, DSP_TPDF_CALC = 7, DSP_TPDF = 8, DSP_WHITE = 9,
    DSP_CLRXY = 10, DSP_SWAPXY = 11, DSP_COPYXY = 12, DSP_COPYYX = 13, DSP_ADDXY = 14,
    DSP_ADDYX = 15, DSP_SUBXY = 16, DSP_SUBYX = 17, DSP_MULXY = 18, DSP_DIVXY = 19,
    DSP_DIVYX = 20, DSP_AVGXY = 21, DSP_AVGYX = 22, DSP_NEGX = 23, DSP_NEGY = 24,
    DSP_SQRTX = 25, DSP_SHIFT = 26, DSP_VALUE = 27, DSP_VALUE_INT = 28, DSP_MUL_VALUE = 29,
    DSP_MUL_VALUE_INT = 30, DSP_DIV_VALUE = 31, DSP_DIV_VALUE_INT = 32, DSP_AND_VALUE_INT = 33,
    DSP_LOAD = 34, DSP_LOAD_GAIN = 35, DSP_LOAD_MUX = 36, DSP_STORE = 37, DSP_LOAD_STORE = 38,
    DSP_LOAD_MEM = 39, DSP_STORE_MEM = 40, DSP_GAIN = 41, DSP_SAT0DB = 42, DSP_SAT0DB_TPDF = 43,
    DSP_SAT0DB_GAIN = 44, DSP_SAT0DB_TPDF_GAIN = 45, DSP_DELAY_1 = 46, DSP_DELAY = 47,
    DSP_DELAY_DP = 48, DSP_DATA_TABLE = 49, DSP_BIQUADS = 50, DSP_FIR = 51, DSP_RMS = 52,
    DSP_DCBLOCK = 53, DSP_DITHER = 54, DSP_DITHER_NS2 = 55, DSP_DISTRIB = 56, DSP_DIRAC = 57,
    DSP_SQUAREWAVE = 58, DSP_CLIP = 59, DSP_LOAD_MEM_DATA = 60, DSP_SINE = 61,
    DSP_MAX_OPCODE = 62
};

/* ---- arithmetic models selected at build time in the reference: dsp_header.h:11-16 ---- */
#define DSP_FORMAT_INT32        1   /* never implemented by the reference (#error)       */
#define DSP_FORMAT_INT64        2   /* int32 samples, int64 accumulator, Q4.28 params     */
#define DSP_FORMAT_FLOAT        3   /* int32 samples, float accumulator                   */
#define DSP_FORMAT_DOUBLE       4   /* int32 samples, double accumulator, float params    */
#define DSP_FORMAT_FLOAT_FLOAT  5   /* float samples, float accumulator                   */
#define DSP_FORMAT_DOUBLE_FLOAT 6   /* float samples, double accumulator, float params    */

/* Q-format mantissa of gains and biquad coefficients in int64 mode: dsp_header.h:258-267 */
#define DSP_MANT   28
#define DSP_MANTBQ 28

/* ---- sample-rate indices: dsp_header.h:136-145; header freqMin/freqMax hold these ---- */
enum dspFreqs {
    F8000, F16000, F24000, F32000, F44100, F48000, F88200, F96000,
    F176400, F192000, F352800, F384000, F705600, F768000, FMAXpos
};
#define DSP_DEFAULT_MIN_FREQ F44100
#define DSP_DEFAULT_MAX_FREQ F192000

/* Hz -> index, FMAXpos when unsupported (dsp_header.h:149-167) */
static inline int dspConvertFrequencyToIndex(int hz)
{
    static const int tab[FMAXpos] = { 8000, 16000, 24000, 32000, 44100, 48000, 88200, 96000,
                                      176400, 192000, 352800, 384000, 705600, 768000 };
    for (int i = 0; i < FMAXpos; i++) if (tab[i] == hz) return i;
    return FMAXpos;
}
/* index -> Hz; anything out of range reads as 768000 (dsp_header.h:169-187) */
static inline int dspConvertFrequencyFromIndex(int idx)
{
    static const int tab[FMAXpos] = { 8000, 16000, 24000, 32000, 44100, 48000, 88200, 96000,
                                      176400, 192000, 352800, 384000, 705600, 768000 };
    return (idx >= 0 && idx < FMAXpos) ? tab[idx] : 768000;
}

/* ---- one program word: dsp_header.h:197-209 ---- */
typedef union opcode_u {
    struct { unsigned short skip; unsigned short opcode; } op;   /* head word                */
    struct { short low; short high; } s16;
    unsigned u32;
    int      i32;
    float    f32;
    int      i[1];
    unsigned u[1];
} opcode_t;

/* ---- program header, 12 words: dsp_header.h:213-228 ---- */
typedef struct dspHeader_s {
    opcode_t head;              /* (DSP_HEADER << 16) | 12                                       */
    int      totalLength;       /* program length in words (even); the data area starts here     */
    int      dataSize;          /* words of run-time state following the program                 */
    unsigned checkSum;          /* sum of all head words (payload words are not summed)          */
    int      numCores;
    int      version;           /* encoder version, 0x102 for the reference at this revision     */
    unsigned short format;      /* mantissa bits (28) when int-encoded, 0 when float-encoded     */
    unsigned short maxOpcode;   /* highest opcode number used                                    */
    int      freqMin;           /* enum dspFreqs index                                           */
    int      freqMax;           /* enum dspFreqs index                                           */
    unsigned usedInputs;        /* bitmap, IO < 32 only                                          */
    unsigned usedOutputs;
    unsigned serialHash;
} dspHeader_t;

#define AVDSP_HEADER_WORDS 12

/*
 * Checksum + core count over a program (dsp_header.h:234-251): walk head words by skip;
 * the END word (skip 0) is not summed; a program without DSP_CORE counts as one core.
 * `limit` bounds the walk (the reference breaks out when the position passes it).
 */
static inline void avdspChecksumWalk(const opcode_t *prog, unsigned limit,
                                     unsigned *sum_out, int *cores_out)
{
    unsigned sum = 0, pos = 0;
    int cores = 0;
    for (;;) {
        unsigned skip = prog[pos].op.skip;
        if (skip == 0) { if (cores == 0) cores = 1; break; }
        if (prog[pos].op.opcode == DSP_CORE) cores++;
        sum += prog[pos].u32;
        pos += skip;
        if (pos > limit) break;
    }
    *sum_out = sum;
    *cores_out = cores;
}

#ifdef __cplusplus
}
#endif
#endif /* AVDSP_FORMAT_H_ */
