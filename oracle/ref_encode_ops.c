/*
 * ref_encode_ops.c -- TEST INFRASTRUCTURE.  Drives the *reference encoder library*
 * (oracle/_ref/libavdspencoder.so, built unmodified by build_ref.sh) through its public API
 * (encoder/dsp_encoder.h:17-223) to emit an "opcode tour": a five-core program that executes every
 * opcode the committed osx/*.bin programs do not reach (X/Y arithmetic, immediates, LOAD_MUX,
 * DELAY_1, DATA_TABLE, TPDF, WHITE, RMS/PWRXY, DCBLOCK, DITHER, DITHER_NS2, DISTRIB, CLIP, ... and, in
 * the float encoding, DIRAC / SQUAREWAVE / SINE).  tests/golden/make_goldens.py runs the result through
 * the compiled reference runtimes to make golden vectors; the oracle (CPU tests) and the general device
 * interpreter (GPU tests) are held to them.
 *
 * DSP_FIR is not in the tour: the reference's dsp_FIR() reads the section header as the first impulse
 * (encoder/dsp_encoder.c:1311-1313), FIR fixtures come from avdsp_amd/progbuilder.py.  DSP_LOAD_MEM_DATA
 * has no encoder function; its two words are appended with addCode() behind a NOP.
 *
 * IO map: inputs IN(k) = 32+k (k = 0..15), outputs 0..31 (windows do not overlap).
 * usage: ref_encode_ops FORMAT OUT.bin        (FORMAT 2 = int64/Q28 encoding, 6 = float encoding)
 */
#include <stdio.h>
#include <stdlib.h>
#include "dsp_encoder.h"
#include "dsp_filters.h"
#include "dsp_fileaccess.h"

#define IN(k) (32 + (k))

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: ref_encode_ops FORMAT OUT.bin\n"); return 2; }
    const int format = atoi(argv[1]);
    const int is_float = format != 2;
    const int fmin = F44100, fmax = F96000, nf = fmax - fmin + 1;
    const int max = 8192;
    opcode_t *buf = (opcode_t *)calloc((size_t)max, sizeof(opcode_t));

    dspEncoderInit(buf, max, format, fmin, fmax, 48);

    /* ---- parameters ---- */
    dsp_PARAM();
    int bank = dspBiquad_Sections(2);
        dsp_Filter2ndOrder(FPEAK, 500.0, 0.9, 1.3f);
        dsp_Filter2ndOrder(FPEAK, 2500.0, 1.1, 0.7f);
    int mux = dspLoadMux_Inputs(3);
        dspLoadMux_Data(IN(0), 0.5);
        dspLoadMux_Data(IN(1), -0.25);
        dspLoadMux_Data(IN(2), 0.125);
    int table;
    if (is_float) {
        float wave[24];
        for (int i = 0; i < 24; i++) wave[i] = (float)(((i * 7) % 24) - 12) / 16.0f;
        table = dspDataTableFloat(wave, 24);
    } else
        table = dspGenerator_Sine(24);
    int mem = dspMem_Location();
    int g1 = dspGain_Default(0.5);
    int val = dspValue_Default(0.75);
    int dly = dspDelay_MicroSec_Max_Default(500, 250);
    float ns2[12];
    for (int f = 0; f < nf; f++) { ns2[3 * f] = 1.0f - 0.05f * f; ns2[3 * f + 1] = -0.5f + 0.02f * f; ns2[3 * f + 2] = 0.25f; }
    int nstab = dspDataTableFloat(ns2, 3 * nf);

    dsp_PARAM_NUM(7);
    int g2 = dspGain_Default(0.9);

    /* ---- core 1: dither source, X/Y arithmetic ---- */
    dsp_CORE();
    int tpdfAddr = dsp_TPDF_CALC(24);
    dsp_STORE(0);
    dsp_WHITE();
    dsp_STORE(1);
    dsp_LOAD_GAIN_Fixed(IN(0), 0.5);
    dsp_COPYXY();
    dsp_LOAD_GAIN_Fixed(IN(1), 0.25);
    dsp_ADDXY();
    dsp_SAT0DB_TPDF();
    dsp_STORE(2);
    dsp_LOAD_GAIN_Fixed(IN(2), 1.0);
    dsp_LOAD_GAIN_Fixed(IN(3), 1.0);
    dsp_SUBXY();
    dsp_SAT0DB();
    dsp_STORE(3);
    dsp_LOAD_GAIN_Fixed(IN(4), 0.5);
    dsp_LOAD_GAIN_Fixed(IN(5), 0.5);
    dsp_ADDYX();
    dsp_NEGX();
    dsp_AVGXY();
    dsp_NEGY();
    dsp_AVGYX();
    dsp_COPYYX();
    dsp_SAT0DB_GAIN(g2);
    dsp_STORE(4);
    dsp_CLRXY();
    dsp_STORE(5);
    dsp_SERIAL(0x1234ABCD);
    dsp_NOP();
    addCode((DSP_LOAD_MEM_DATA << 16) | 2);         /* X = the value TPDF_CALC left in the data area */
    addCode(tpdfAddr);
    dsp_STORE(28);

    /* ---- core 2: immediates, multiply / divide, shift, sqrt, and, clip ---- */
    dsp_CORE();
    dsp_LOAD(IN(0));
    dsp_VALUE_FixedInt(3);
    dsp_SWAPXY();
    dsp_DIVXY();
    dsp_MUL_FixedInt(2);
    dsp_DIV_FixedInt(5);
    dsp_STORE(6);
    dsp_LOAD(IN(1));
    dsp_MUL_Fixed(0.5);
    dsp_DIV_Fixed(0.25);
    dsp_SHIFT(-1);
    dsp_STORE(7);
    dsp_VALUE_FixedInt(1000);
    dsp_LOAD(IN(2));
    dsp_MULXY();
    dsp_DIV_FixedInt(1000);
    dsp_STORE(8);
    dsp_VALUE_FixedInt(7);
    dsp_LOAD(IN(6));
    dsp_SWAPXY();                                    /* X = 7, Y = sample */
    dsp_DIVYX();                                     /* Y = sample / 7 */
    dsp_SWAPXY();
    dsp_STORE(29);
    dsp_LOAD(IN(3));
    dsp_COPYXY();
    dsp_MULXY();
    dsp_SQRTX();
    dsp_STORE(9);
    dsp_LOAD(IN(4));
    dsp_AND_FixedInt(0x00FFFF00);
    dsp_STORE(10);
    dsp_LOAD_GAIN_Fixed(IN(5), 1.0);
    dsp_CLIP_Fixed(0.25);
    dsp_SAT0DB();
    dsp_STORE(11);
    dsp_VALUE(val);
    dsp_GAIN(g1);
    dsp_SHIFT(3);
    dsp_STORE_MEM(mem);
    dsp_VALUE_Fixed(-0.3);
    dsp_GAIN_Fixed(0.5);
    dsp_SAT0DB_TPDF_GAIN_Fixed(1.5);
    dsp_STORE(12);

    /* ---- core 3: mixer, memories, delay lines, table, local dither depth ---- */
    dsp_CORE();
    dsp_LOAD_MUX(mux);
    dsp_DELAY_1();
    dsp_SAT0DB();
    dsp_STORE(13);
    dsp_DATA_TABLE(table, 0.5, 3, 24);
    if (!is_float) { dsp_SAT0DB(); }
    dsp_STORE(14);
    dsp_LOAD(IN(6));
    dsp_DELAY(dly);
    dsp_STORE(15);
    dsp_LOAD_GAIN_Fixed(IN(7), 1.0);
    dsp_DELAY_DP_FixedMicroSec(100);
    dsp_BIQUADS(bank);
    dsp_SAT0DB_TPDF_GAIN_Fixed(0.8);
    dsp_STORE(16);
    dsp_LOAD_MEM(mem);
    dsp_SAT0DB();
    dsp_STORE(30);
    dsp_TPDF(20);
    dsp_LOAD_GAIN_Fixed(IN(8), 0.7);
    dsp_SAT0DB_TPDF();
    dsp_STORE(17);

    /* ---- core 4: meters, DC blocker, noise-shaped dither, histogram ---- */
    dsp_CORE();
    dsp_LOAD(IN(9));
    dsp_RMS(10, 2);
    dsp_STORE(18);
    dsp_LOAD(IN(10));
    dsp_LOAD(IN(11));
    dsp_PWRXY(10, 0);
    dsp_STORE(19);
    dsp_LOAD_GAIN_Fixed(IN(12), 1.0);
    dsp_DCBLOCK(10);
    dsp_SAT0DB();
    dsp_STORE(20);
    dsp_LOAD_GAIN_Fixed(IN(13), 0.5);
    dsp_DITHER();
    dsp_SAT0DB();
    dsp_STORE(21);
    dsp_LOAD_GAIN_Fixed(IN(14), 0.5);
    dsp_DITHER_NS2(nstab);
    dsp_SAT0DB();
    dsp_STORE(22);
    dsp_LOAD(IN(15));
    dsp_DISTRIB(23, 16);

    /* ---- core 5: generators (their int64 bodies call functions the reference never defines) ---- */
    if (is_float) {
        dsp_CORE();
        dsp_DIRAC_Fixed(1000, 0.5);
        dsp_STORE(24);
        dsp_SQUAREWAVE_Fixed(500, 0.5);
        dsp_STORE(25);
        dsp_SINE_Fixed(1000, 0.5);
        dsp_STORE(26);
        dsp_SWAPXY();
        dsp_STORE(27);
    }

    int size = dsp_END_OF_CODE();
    if (dspCreateBuffer(argv[2], (int *)buf, size) != size) { fprintf(stderr, "write failed\n"); return 1; }
    printf("words=%d data=%d\n", size, dspHeaderPtr->dataSize);
    return 0;
}
