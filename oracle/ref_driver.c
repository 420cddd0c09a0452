/*
 * ref_driver.c -- TEST INFRASTRUCTURE.  Runs an encoded program through a *reference* runtime
 * library (oracle/_ref/libavdspref_N.so, built by build_ref.sh) exactly the way the reference
 * hosts do (linux/avdsp_plugin.c:95-142, linux/dsprun.c:147-171: cores outer, frames inner,
 * frame-interleaved 32-bit samples) and writes the raw result.  tests/golden/make_goldens.py
 * calls it to produce the golden vectors; nothing in the product path knows it exists.
 *
 * The library is opened with RTLD_LAZY so that the int build's two never-defined,
 * never-called symbols (dspQNMmax, DSP_Q31) stay unbound instead of being stubbed.
 *
 * usage: ref_driver LIB FORMAT PROG.bin MAXSIZE FS RANDOM DITHER IN.raw OUT.raw NFRAMES BLOCK
 *                   IN_STRIDE IN_IO_BASE OUT_STRIDE OUT_IO_BASE SCRATCH_LEN [STATE_OUT.raw|- [REPEAT [WARM]]]
 *   MAXSIZE 0 = totalLength + dataSize.  Prints "init=<rc>" and, on success, "cores=<n>".
 *   REPEAT > 1 runs the NFRAMES input that many times back to back (timing runs for bench.py's
 *   cpu_baseline leg); the first WARM passes are excluded from the "elapsed=<s> frames=<n>" line.
 *   A 21st argument PRIMER.bin is a program initialised (at FS) BEFORE the real one, in the same process: the reference
 *   keeps its rate count in a static that dspRuntimeInit's format conversion reads before dspRuntimeReset refreshes it
 *   (dsp_runtime.c:106,131,181-190), so what a load converts depends on what was loaded before.
 */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "avdsp_format.h"

typedef int       (*init_fn)(opcode_t *, int, int, int, int);
typedef opcode_t *(*find_fn)(opcode_t *, int);
typedef opcode_t *(*begin_fn)(opcode_t *);
typedef int       (*run_fn)(opcode_t *, int *, void *);

static void *slurp(const char *path, size_t *bytes)
{
    FILE *f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    void *p = malloc(n > 0 ? (size_t)n : 1);
    if (n > 0 && fread(p, 1, (size_t)n, f) != (size_t)n) { perror("fread"); exit(2); }
    fclose(f);
    *bytes = (size_t)n;
    return p;
}

int main(int argc, char **argv)
{
    if (argc < 17) { fprintf(stderr, "ref_driver: bad arguments\n"); return 2; }
    const char *lib = argv[1];
    int format = atoi(argv[2]);
    const char *progpath = argv[3];
    int maxsize = atoi(argv[4]), fs = atoi(argv[5]), rnd = atoi(argv[6]), dither = atoi(argv[7]);
    const char *inpath = argv[8], *outpath = argv[9];
    int nframes = atoi(argv[10]), block = atoi(argv[11]);
    int in_stride = atoi(argv[12]), in_base = atoi(argv[13]);
    int out_stride = atoi(argv[14]), out_base = atoi(argv[15]);
    int scratch_len = atoi(argv[16]);
    const char *statepath = (argc > 17 && strcmp(argv[17], "-")) ? argv[17] : NULL;
    int repeat = argc > 18 ? atoi(argv[18]) : 1, warm = argc > 19 ? atoi(argv[19]) : 0;
    struct timespec ts0, ts1;

    void *h = dlopen(lib, RTLD_LAZY | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    char name[32];
    snprintf(name, sizeof name, "dspRuntime_%d", format);
    init_fn  f_init  = (init_fn)dlsym(h, "dspRuntimeInit");
    find_fn  f_find  = (find_fn)dlsym(h, "dspFindCore");
    begin_fn f_begin = (begin_fn)dlsym(h, "dspFindCoreBegin");
    run_fn   f_run   = (run_fn)dlsym(h, name);
    if (!f_init || !f_find || !f_begin || !f_run) { fprintf(stderr, "missing symbol\n"); return 2; }

    if (argc > 20) {                                   /* prime the reference's statics with another program first */
        size_t qb;
        void *q = slurp(argv[20], &qb);
        const dspHeader_t *qh = (const dspHeader_t *)q;
        size_t qn = (size_t)qh->totalLength + (size_t)(qh->dataSize > 0 ? qh->dataSize : 0);
        if (qn < qb / 4) qn = qb / 4;
        opcode_t *qc = (opcode_t *)calloc(qn + 64, 4);
        memcpy(qc, q, qb);
        printf("primer_init=%d\n", f_init(qc, (int)qn, fs, rnd, dither));
    }
    size_t pbytes;
    void *raw = slurp(progpath, &pbytes);
    const dspHeader_t *hd = (const dspHeader_t *)raw;
    size_t need = (size_t)hd->totalLength + (size_t)(hd->dataSize > 0 ? hd->dataSize : 0);
    if (need < pbytes / 4) need = pbytes / 4;
    opcode_t *code = (opcode_t *)calloc(need + 64, 4);
    memcpy(code, raw, pbytes);
    if (maxsize == 0) maxsize = (int)need;

    int rc = f_init(code, maxsize, fs, rnd, dither);
    printf("init=%d\n", rc);
    if (rc < 0) return 0;
    int *data = (int *)code + rc;

    opcode_t *cores[64];
    int ncores = 0;
    for (; ncores < 64; ncores++) {
        opcode_t *p = f_find(code, ncores + 1);
        if (!p) break;
        cores[ncores] = f_begin(p);
    }
    printf("cores=%d\n", ncores);

    size_t ibytes;
    unsigned *in = (unsigned *)slurp(inpath, &ibytes);
    if (ibytes < (size_t)nframes * in_stride * 4) { fprintf(stderr, "input too short\n"); return 2; }
    unsigned *out = (unsigned *)calloc((size_t)nframes * out_stride + 1, 4);
    unsigned *scratch = (unsigned *)calloc((size_t)scratch_len + 1, 4);
    if (block <= 0) block = nframes;

    if (repeat < 1) repeat = 1;
    clock_gettime(CLOCK_MONOTONIC, &ts0);
    for (int rep = 0; rep < repeat; rep++) {
    if (rep == warm) clock_gettime(CLOCK_MONOTONIC, &ts0);
    for (int b0 = 0; b0 < nframes; b0 += block) {
        int b1 = b0 + block < nframes ? b0 + block : nframes;
        for (int nc = 0; nc < ncores; nc++)
            for (int n = b0; n < b1; n++) {
                memcpy(scratch + out_base, out + (size_t)n * out_stride, (size_t)out_stride * 4);
                memcpy(scratch + in_base,  in  + (size_t)n * in_stride,  (size_t)in_stride * 4);
                f_run(cores[nc], data, scratch);
                memcpy(out + (size_t)n * out_stride, scratch + out_base, (size_t)out_stride * 4);
            }
    }
    }
    clock_gettime(CLOCK_MONOTONIC, &ts1);
    if (repeat > 1)
        printf("elapsed=%.6f frames=%ld\n", (ts1.tv_sec - ts0.tv_sec) + 1e-9 * (ts1.tv_nsec - ts0.tv_nsec),
               (long)(repeat - warm) * nframes);

    FILE *f = fopen(outpath, "wb");
    if (!f) { perror(outpath); return 2; }
    fwrite(out, 4, (size_t)nframes * out_stride, f);
    fclose(f);
    if (statepath) {
        f = fopen(statepath, "wb");
        if (!f) { perror(statepath); return 2; }
        fwrite(code, 4, (size_t)rc + (size_t)hd->dataSize, f);     /* program (possibly mutated) + state */
        fclose(f);
    }
    return 0;
}
