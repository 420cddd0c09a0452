/*
 * enc_sweep.c -- TEST INFRASTRUCTURE.  An encoder workout written against the public encoder API
 * (encoder/dsp_encoder.h:17-223, dsp_filters.h:30-76): every filter type and preset at assorted
 * frequencies / Q / gains, sine tables of several sizes, delay parameters in microseconds and
 * millimetres, RMS / PWRXY in both time bases, PARAM_NUM, inline parameters, memories, mux lists with
 * flexible counts ...  The program is not meant to sound like anything.  build_ref.sh links it with the
 * reference encoder (oracle/_ref/enc_sweep) and tests/golden/make_goldens.py commits what that writes as
 * tests/golden/enc_sweep_<variant>.bin; tests/test_encoder.py links the same file with
 * avdsp_amd/lib/libavdsp_encoder.so and requires identical bytes.
 *
 * usage: enc_sweep FORMAT FMIN_INDEX FMAX_INDEX OUT.bin
 */
#include <stdio.h>
#include <stdlib.h>
#include "dsp_encoder.h"
#include "dsp_filters.h"
#include "dsp_fileaccess.h"

typedef int (*preset_fn)(dspFilterParam_t);

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: enc_sweep FORMAT FMIN FMAX OUT.bin\n"); return 2; }
    const int format = atoi(argv[1]), fmin = atoi(argv[2]), fmax = atoi(argv[3]);
    const int is_float = format != 2;
    const int max = 60000;
    opcode_t *buf = (opcode_t *)calloc((size_t)max, sizeof(opcode_t));

    dspEncoderInit(buf, max, format, fmin, fmax, 32);

    static const preset_fn presets[] = {
        dsp_LP_BES2, dsp_HP_BES2, dsp_LP_BES2_3DB, dsp_HP_BES2_3DB, dsp_LP_BUT2, dsp_HP_BUT2, dsp_LP_LR2, dsp_HP_LR2,
        dsp_LP_BES3, dsp_HP_BES3, dsp_LP_BES3_3DB, dsp_HP_BES3_3DB, dsp_LP_BUT3, dsp_HP_BUT3, dsp_LP_LR3, dsp_HP_LR3,
        dsp_LP_BES4, dsp_HP_BES4, dsp_LP_BES4_3DB, dsp_HP_BES4_3DB, dsp_LP_BUT4, dsp_HP_BUT4, dsp_LP_LR4, dsp_HP_LR4,
        dsp_LP_BES6, dsp_HP_BES6, dsp_LP_BES6_3DB, dsp_HP_BES6_3DB, dsp_LP_BUT6, dsp_HP_BUT6, dsp_LP_LR6, dsp_HP_LR6,
        dsp_LP_BES8, dsp_HP_BES8, dsp_HP_BES8_3DB, dsp_LP_BUT8, dsp_HP_BUT8, dsp_LP_LR8, dsp_HP_LR8
    };
    static const double freqs[] = { 20.0, 63.5, 250.0, 1000.0, 3150.0, 7999.0, 15000.0 };
    enum { NPRE = sizeof presets / sizeof presets[0], NFREQ = sizeof freqs / sizeof freqs[0] };

    /* ---- region 1: presets, each in its own flexible bank ---- */
    dsp_PARAM();
    int banks[NPRE];
    for (int p = 0; p < NPRE; p++) {
        banks[p] = dspBiquad_Sections_Flexible();
        presets[p](freqs[p % NFREQ] * (1.0 + 0.01 * p));
    }

    /* ---- region 2: every basic type over frequency / Q / gain, banks with exact and maximum counts ---- */
    dsp_PARAM_NUM(3);
    static const int second[] = { FLP2, FHP2, FLS2, FHS2, FAP2, FPEAK, FNOTCH, FBP0DB, FBPQ };
    static const int first[]  = { FLP1, FHP1, FLS1, FHS1, FAP1 };
    static const double qs[] = { 0.3, 0.70710678, 1.0, 4.5, 0.0 };
    static const float gains[] = { 1.0f, 0.25f, 1.9952623f, 3.5f };
    int bank2 = dspBiquad_Sections(9 * 5);
    for (int t = 0; t < 9; t++)
        for (int k = 0; k < 5; k++)
            dsp_Filter2ndOrder(second[t], freqs[(t + k) % NFREQ], qs[k], gains[(t + k) % 4]);
    int bank1 = dspBiquad_Sections_Maximum(40);
    for (int t = 0; t < 5; t++)
        for (int k = 0; k < 4; k++)
            dsp_Filter1stOrder(first[t], freqs[(t + 2 * k) % NFREQ], gains[k]);
    int bankg = dspBiquad_Sections(4);
    dsp_filter(LPBE3db4, 440.0, 0.0, 1.0);                       /* 2 cells */
    dsp_filter(FPEAK, 880.0, 2.0, 1.5f);
    dsp_filter(FHS1, 5000.0, 0.0, 0.5f);

    /* ---- region 3: tables, delays, memories, mixer ---- */
    dsp_PARAM();
    int sines[5];
    static const int sizes[5] = { 4, 24, 100, 192, 1024 };
    for (int i = 0; i < 5; i++) sines[i] = dspGenerator_Sine(sizes[i]);
    int d1 = dspDelay_MicroSec_Max(1000);
    int d2 = dspDelay_MicroSec_Max_Default(2000, 333);
    int d3 = dspDelay_MilliMeter_Max(300, 340.0f);
    int d4 = dspDelay_MilliMeter_Max_Default(500, 123, 343.5f);
    int mems = dspMem_LocationMultiple(3);
    int mux = dspLoadMux_Inputs(0);
        dspLoadMux_Data(1, 0.5f); dspLoadMux_Data(2, -0.5f); dspLoadMux_Data(3, 1.9990234f); dspLoadMux_Data(4, -2.0f);
    int g = dspGain_Default(0.70794576f);
    int v = dspValue_Default(-1.25f);
    int pair = dspData2(17, -4);
    int quad = dspData4(1, 2, 3, 4);
    int six = dspData6(-1, -2, -3, -4, -5, -6);
    int eight = dspData8(10, 20, 30, 40, 50, 60, 70, 80);
    int itab[7] = { 5, -5, 100000, -100000, 0x7FFFFFFF, (int)0x80000000, 0 };
    int it = dspDataTableInt(itab, 7);
    float ftab[9] = { 0.0f, 1.0f, -1.0f, 0.333333f, 7.9f, -7.9f, 1e-6f, 12.0f, -12.0f };   /* the last two saturate Q4.28 */
    int ft = dspDataTableFloat(ftab, 9);
    int ns2 = 0;
    if (fmin >= F44100 && fmax <= F192000) {                     /* the only range dsp_DITHER_NS2 accepts */
        float c[3 * 6];
        for (int i = 0; i < 3 * (fmax - fmin + 1); i++) c[i] = (float)(((i * 5) % 7) - 3) / 4.0f;
        ns2 = dspDataTableFloat(c, 3 * (fmax - fmin + 1));
    }
    (void)pair; (void)quad; (void)six; (void)eight; (void)it; (void)bank1; (void)bankg;

    /* ---- core 1 ---- */
    dsp_CORE();
    dsp_TPDF_CALC(0);
    for (int p = 0; p < NPRE; p++) {
        dsp_LOAD_GAIN(p % 8, g);
        dsp_BIQUADS(banks[p]);
        dsp_SAT0DB_TPDF_GAIN(g);
        dsp_STORE(8 + p % 8);
    }
    dsp_LOAD_MUX(mux);
    dsp_BIQUADS(bank2);
    dsp_GAIN(g);
    dsp_SAT0DB_GAIN(g);
    dsp_STORE(16);
    dsp_LOAD_STORE();
        dspLoadStore_Data(0, 17); dspLoadStore_Data(1, 18); dspLoadStore_Data(7, 31);

    /* ---- core 2 ---- */
    dsp_CORE();
    dsp_LOAD(3);
    dsp_DELAY(d1);
    dsp_DELAY(d2);
    dsp_DELAY_DP(d3);
    dsp_DELAY_DP(d4);
    dsp_DELAY_FixedMicroSec(20);
    dsp_DELAY_FixedMilliMeter(170, 340.0f);
    dsp_DELAY_DP_FixedMicroSec(1);
    dsp_DELAY_DP_FixedMilliMeter(33, 330.0f);
    dsp_DELAY_1();
    dsp_STORE_MEM(mems);
    dsp_STORE_MEM_Index(mems, 2);
    dsp_LOAD_MEM_Index(mems, 1);
    dsp_VALUE(v);
    dsp_VALUE_Fixed(3.25f);
    dsp_VALUE_FixedInt(-77);
    dsp_GAIN_Fixed(dB2gain(-6.0f));
    dsp_MUL_Fixed(1.5f); dsp_DIV_Fixed(0.75f); dsp_MUL_FixedInt(3); dsp_DIV_FixedInt(-9); dsp_AND_FixedInt(0x0FF0);
    dsp_SHIFT(5); dsp_SHIFT_FixedInt(-7);
    for (int i = 0; i < 5; i++) dsp_DATA_TABLE(sines[i], 0.9f, 1 + i, sizes[i]);
    dsp_DATA_TABLE(ft, 1.0f, 2, 9);
    dsp_DATA_TABLE(0, 0.5f, 1, 4);
        addCode(1); addCode(2); addCode(3); addCode(4);
    dsp_STORE(19);
    dsp_NOP();
    opcodeIndexAligned8(); opcodeIndexMisAligned8();
    dsp_SERIAL(0xDEADBEEF);

    /* ---- core 3 ---- */
    dsp_CORE();
    dsp_TPDF(17);
    dsp_LOAD(5);
    dsp_RMS(100, 10);
    dsp_RMS(10, 0);
    dsp_RMS_MilliSec(1000, 50);
    dsp_RMS_MilliSec(300, 0);
    dsp_PWRXY(50, 3);
    dsp_PWRXY_MilliSec(400, 100);
    dsp_DCBLOCK(1); dsp_DCBLOCK(37); dsp_DCBLOCK(100);
    dsp_DITHER();
    dsp_DISTRIB(20, 8); dsp_DISTRIB(21, 1024);
    dsp_CLIP_Fixed(0.999f); dsp_CLIP_Fixed(-0.5f);
    if (ns2) dsp_DITHER_NS2(ns2);
    dsp_SAT0DB(); dsp_SAT0DB_TPDF(); dsp_SAT0DB_GAIN_Fixed(0.1f); dsp_SAT0DB_TPDF_GAIN_Fixed(7.5f);
    dsp_WHITE(); dsp_CLRXY(); dsp_SWAPXY(); dsp_COPYXY(); dsp_COPYYX();
    dsp_ADDXY(); dsp_ADDYX(); dsp_SUBXY(); dsp_SUBYX(); dsp_MULXY(); dsp_DIVXY(); dsp_DIVYX();
    dsp_AVGXY(); dsp_AVGYX(); dsp_SQRTX(); dsp_NEGX(); dsp_NEGY();
    dsp_STORE(22);
    if (is_float) {
        dsp_DIRAC_Fixed(100, 0.9f); dsp_SQUAREWAVE_Fixed(3000, 0.1f);
        dsp_SINE_Fixed(20, 1.0f); dsp_SINE_Fixed(1999, 0.001f);
        dsp_STORE(23);
    }
    setSerialHash(0xCAFE0001u);

    int size = dsp_END_OF_CODE();
    if (dspCreateBuffer(argv[4], (int *)buf, size) != size) { fprintf(stderr, "write failed\n"); return 1; }
    printf("words=%d data=%d\n", size, dspHeaderPtr->dataSize);
    return 0;
}
