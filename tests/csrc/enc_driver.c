/*
 * enc_driver.c -- test helper: the few lines of the reference's `dspcreate` host that matter for the
 * program words (encoder/dspcreate.c:16-21,92-119,147-158): parse -dspformat / -fsmin / -fsmax, set up
 * the encoder with a 10000-word table and 32 IOs, hand the remaining arguments to the program's
 * dspProg(), write the words.  Linked by tests/test_encoder.py with a DSP program source and
 * avdsp_amd/lib/libavdsp_encoder.so.
 *
 * usage: enc_driver OUT.bin [-dspformat N] [-fsmin HZ] [-fsmax HZ] [program arguments ...]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "avdsp_encoder.h"

extern int dspProg(int argc, char **argv);

int main(int argc, char **argv)
{
    static opcode_t table[10000];
    int format = DSP_FORMAT_FLOAT, fmin = DSP_DEFAULT_MIN_FREQ, fmax = DSP_DEFAULT_MAX_FREQ, i;
    if (argc < 2) return 2;
    for (i = 2; i < argc; i++) {
        if (!strcmp(argv[i], "-dspformat") && i + 1 < argc) { format = atoi(argv[++i]); continue; }
        if (!strcmp(argv[i], "-fsmin") && i + 1 < argc) { fmin = dspConvertFrequencyToIndex(atoi(argv[++i])); continue; }
        if (!strcmp(argv[i], "-fsmax") && i + 1 < argc) { fmax = dspConvertFrequencyToIndex(atoi(argv[++i])); continue; }
        break;
    }
    if (fmin >= FMAXpos || fmax >= FMAXpos) return 2;
    dspEncoderInit(table, 10000, format, fmin, fmax, 32);
    int size = dspProg(argc - i, &argv[i]);
    if (size <= 0 || dspCreateBuffer(argv[1], (int *)table, size) != size) return 1;
    return 0;
}
