/*
 * ref_encode.c -- TEST INFRASTRUCTURE.  Drives the *reference encoder library* (compiled unmodified
 * by oracle/build_ref.sh into oracle/_ref/libavdspencoder.so) through its public API
 * (encoder/dsp_encoder.h:17-223, encoder/dsp_filters.h:30-76) to emit the synthetic
 * N-channel biquad-cascade program of SURVEY.md section 8(d).  tests/golden/make_goldens.py compares
 * the result byte for byte with avdsp_amd/progbuilder.py, which restates the same layout.
 *
 *   channel c:  PARAM{ bank of S peaking-EQ sections }  LOAD_GAIN(IO=C+c, 1.0)  BIQUADS(bank)
 *               SAT0DB  STORE(IO=c)        one CORE in front, END_OF_CODE behind.
 *   section b of channel c: FPEAK f0 = 100+37b+3(c mod 97) Hz, Q = 0.7+0.05(b mod 5),
 *               gain 1.2 (odd b) / 0.8 (even b).
 *
 * usage: ref_encode FORMAT C S FMIN_INDEX FMAX_INDEX OUT.bin      (FORMAT: 2 = int64/Q28, 4|6 = float)
 * FIR programs are NOT generated here: the reference's dsp_FIR() emits a wrong impulse pointer
 * (encoder/dsp_encoder.c:1311-1313), so FIR fixtures come from progbuilder.py alone.
 */
#include <stdio.h>
#include <stdlib.h>
#include "dsp_encoder.h"
#include "dsp_filters.h"
#include "dsp_fileaccess.h"

int main(int argc, char **argv)
{
    if (argc < 7) { fprintf(stderr, "usage: ref_encode FORMAT C S FMIN FMAX OUT.bin\n"); return 2; }
    int format = atoi(argv[1]), C = atoi(argv[2]), S = atoi(argv[3]);
    int fmin = atoi(argv[4]), fmax = atoi(argv[5]);
    int nf = fmax - fmin + 1;
    long max = 64 + (long)C * (16 + (long)S * (2 + 6 * nf) + 8);
    opcode_t *buf = (opcode_t *)calloc((size_t)max, sizeof(opcode_t));

    dspEncoderInit(buf, (int)max, format, fmin, fmax, 2 * C);
    dsp_CORE();
    for (int c = 0; c < C; c++) {
        dsp_PARAM();
        int bank = dspBiquad_Sections(S);
        for (int b = 0; b < S; b++) {
            double f0 = 100.0 + 37.0 * b + 3.0 * (c % 97);
            double Q = 0.7 + 0.05 * (b % 5);
            float gain = (b & 1) ? 1.2f : 0.8f;
            dsp_Filter2ndOrder(FPEAK, f0, Q, gain);
        }
        dsp_LOAD_GAIN_Fixed(C + c, 1.0);
        dsp_BIQUADS(bank);
        dsp_SAT0DB();
        dsp_STORE(c);
    }
    int size = dsp_END_OF_CODE();
    if (dspCreateBuffer(argv[6], (int *)buf, size) != size) { fprintf(stderr, "write failed\n"); return 1; }
    printf("words=%d data=%d\n", size, dspHeaderPtr->dataSize);
    return 0;
}
