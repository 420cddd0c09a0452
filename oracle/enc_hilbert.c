/*
 * enc_hilbert.c -- TEST INFRASTRUCTURE.  dsp_Hilbert() (encoder/dsp_filters.h:76, dsp_filters.c:218-240, design in
 * encoder/dsp_HilbertDesign.c) through the public encoder API: both branches of the phase splitter for 1 .. 10 stages at
 * assorted transition widths, then a core per branch pair so that the file is a runnable program.  build_ref.sh links it with
 * the REFERENCE encoder (oracle/_ref/enc_hilbert) and tests/golden/make_hilbert_goldens.py commits what that writes as
 * tests/golden/enc_hilbert_<variant>.bin; tests/test_encoder.py links the same file with avdsp_amd/lib/libavdsp_encoder.so
 * and requires identical bytes.  The reference's own use: dspprogs/oktodac_fabriceo.c:198-201 (4 stages, 160 Hz).
 *
 * usage: enc_hilbert FORMAT FMIN_INDEX FMAX_INDEX OUT.bin
 */
#include <stdio.h>
#include <stdlib.h>
#include "dsp_encoder.h"
#include "dsp_filters.h"
#include "dsp_fileaccess.h"

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: enc_hilbert FORMAT FMIN FMAX OUT.bin\n"); return 2; }
    const int format = atoi(argv[1]), fmin = atoi(argv[2]), fmax = atoi(argv[3]);
    const int max = 60000;
    opcode_t *buf = (opcode_t *)calloc((size_t)max, sizeof(opcode_t));
    static const double widths[] = { 160.0, 120.0, 40.0, 20.0, 500.0, 1000.0, 77.7, 250.0, 2000.0, 10.0 };

    dspEncoderInit(buf, max, format, fmin, fmax, 32);
    dsp_PARAM();
    int ref[10], quad[10];
    for (int st = 1; st <= 10; st++) {
        ref[st - 1] = dspBiquad_Sections(st);
        dsp_Hilbert(st, widths[st - 1], 0);
        quad[st - 1] = dspBiquad_Sections(st);
        dsp_Hilbert(st, widths[st - 1], 90);
    }
    /* the reference's own call (dspprogs/oktodac_fabriceo.c:198-201) in flexible banks */
    int fref = dspBiquad_Sections_Flexible();
    dsp_Hilbert(4, 160.0, 0);
    int fquad = dspBiquad_Sections_Flexible();
    dsp_Hilbert(4, 160.0, 90);

    /* one core: input 24 + (st mod 2) through both branches of every pair -> outputs 0 .. 21 */
    dsp_CORE();
    for (int st = 1; st <= 10; st++) {
        dsp_LOAD_GAIN_Fixed(24 + (st & 1), 0.5);
        dsp_BIQUADS(ref[st - 1]);
        dsp_SAT0DB();
        dsp_STORE(2 * (st - 1));
        dsp_LOAD_GAIN_Fixed(24 + (st & 1), 0.5);
        dsp_BIQUADS(quad[st - 1]);
        dsp_SAT0DB();
        dsp_STORE(2 * (st - 1) + 1);
    }
    dsp_LOAD_GAIN_Fixed(24, 0.5); dsp_BIQUADS(fref);  dsp_SAT0DB(); dsp_STORE(20);
    dsp_LOAD_GAIN_Fixed(24, 0.5); dsp_BIQUADS(fquad); dsp_SAT0DB(); dsp_STORE(21);

    int size = dsp_END_OF_CODE();
    if (dspCreateBuffer(argv[4], (int *)buf, size) != size) { fprintf(stderr, "write failed\n"); return 1; }
    printf("words=%d data=%d\n", size, dspHeaderPtr->dataSize);
    return 0;
}
