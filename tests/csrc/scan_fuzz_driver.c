/*
 * scan_fuzz_driver.c -- TEST INFRASTRUCTURE.  Memory-safety cross-check of the host's opcode scan
 * (avdsp_host.c scan_generic / lower_core, reached host-only through dspRuntimeCoreInfo): a program the
 * scan ACCEPTS must be one the interpreter can run without touching a word outside the caller's buffer or
 * the samples[] frame.  The device interpreter follows the same offsets as the oracle's, so the oracle,
 * built with AddressSanitizer, stands in for it here: the program buffer and the frame are exact-size heap
 * blocks, any stray access aborts the run.
 *
 *   scan_fuzz_driver PROGRAM.bin FORMAT FS NFRAMES  ->  exit 0: refused or ran clean; ASan abort otherwise
 * stdout: "refused <code>" | "ran <ncores>"
 * Built by tests/test_scan_fuzz.py: gcc -fsanitize=address,undefined, linked with libavdsp_mi355x.so.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "avdsp_runtime.h"
#include "../../oracle/avdsp_oracle.h"

#define FRAME_WORDS 65536         /* the scan accepts IO numbers below 65536; the device frame has max(IO span, 4096) words */

int main(int argc, char **argv)
{
    if (argc < 5) return 2;
    const int format = atoi(argv[2]), fs = atoi(argv[3]), nframes = atoi(argv[4]);
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    int nwords = (int)(bytes / 4);
    if (nwords < 12) return 2;
    int *file = (int *)malloc((size_t)nwords * 4);
    if (fread(file, 4, (size_t)nwords, f) != (size_t)nwords) return 2;
    fclose(f);
    /* header: word 1 = totalLength, word 2 = dataSize (dsp_header.h:216-228) */
    long long total = (long long)file[1] + (long long)file[2];
    if (file[1] < 12 || file[1] > nwords || file[2] < 0 || total > (1 << 24)) { printf("refused header\n"); return 0; }
    int *buf = (int *)calloc((size_t)total, 4);               /* exact size: ASan guards both ends */
    int *buf2 = (int *)calloc((size_t)total, 4);
    memcpy(buf, file, (size_t)file[1] * 4);
    memcpy(buf2, file, (size_t)file[1] * 4);

    int rc = dspRuntimeInit((opcode_t *)buf, (int)total, fs, 7, 24);
    if (rc < 0) { printf("refused init %d\n", rc); return 0; }
    opcode_t *cores[64];
    int ncores = 0;
    for (int k = 1; k <= 64; k++) {
        opcode_t *c = dspFindCore((opcode_t *)buf, k);
        if (!c) break;
        cores[ncores++] = dspFindCoreBegin(c);
    }
    for (int k = 0; k < ncores; k++) {
        int a, b, c;
        rc = dspRuntimeCoreInfo(format, cores[k], &a, &b, &c);
        if (rc < 0) { printf("refused core %d: %d %s\n", k, rc, dspRuntimeLastError()); return 0; }
    }
    /* every core accepted: run them in the oracle on its own copy */
    oracle_ctx *o = oracle_new(format);
    rc = oracle_init(o, (opcode_t *)buf2, (int)total, fs, 7, 24);
    if (rc < 0) { printf("oracle refuses %d\n", rc); return 0; }
    unsigned *frame = (unsigned *)calloc(FRAME_WORDS, 4);
    unsigned s = 12345;
    for (int n = 0; n < nframes; n++) {
        for (int k = 0; k < 64; k++) {
            s = s * 1664525u + 1013904223u;
            frame[k] = (format == 5 || format == 6) ? 0x3c000000u + (s >> 12) : (unsigned)((int)s >> 3);
        }
        for (int k = 0; k < ncores; k++) {
            opcode_t *c = (opcode_t *)buf2 + (cores[k] - (opcode_t *)buf);
            oracle_run(o, c, buf2 + rc, frame);
        }
    }
    printf("ran %d\n", ncores);
    oracle_free(o);
    free(frame); free(buf); free(buf2); free(file);
    return 0;
}
