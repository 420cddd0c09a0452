/*
 * host_demo.c -- a C host written against the REFERENCE's API only plus the block extension, the way
 * linux/avdsp_plugin.c and linux/dsprun.c use the runtime: read an encoded program, dspRuntimeInit,
 * find the cores, push interleaved S32 frames through every core in blocks, write the result.
 * Link with -lavdsp_mi355x instead of compiling runtime/dsp_runtime.c in:
 *
 *   gcc -Iinclude -DDSP_FORMAT=2 examples/host_demo.c -Lavdsp_amd/lib -lavdsp_mi355x \
 *       -Wl,-rpath,$PWD/avdsp_amd/lib -o host_demo
 *   ./host_demo prog.bin 48000 in.raw nbchin in_io_base out.raw nbchout out_io_base [block]
 *
 * DSP_FORMAT selects the entry points exactly like the reference's DSP_RUNTIME_FORMAT() macro does.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "compat/dsp_runtime.h"          /* the reference's names: DSP_RUNTIME_FORMAT(), dspSample_t (dsp_runtime.h:24-131) */

#define OPCODES_MAX 20000
#define CORES_MAX 8

int main(int argc, char **argv)
{
    if (argc < 9) { fprintf(stderr, "usage: host_demo prog.bin fs in.raw nbchin in_io_base out.raw nbchout out_io_base [block]\n"); return 2; }
    const int fs = atoi(argv[2]), nbchin = atoi(argv[4]), in_base = atoi(argv[5]);
    const int nbchout = atoi(argv[7]), out_base = atoi(argv[8]), block = argc > 9 ? atoi(argv[9]) : 256;

    static opcode_t opcodes[OPCODES_MAX];
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    size_t words = fread(opcodes, 4, OPCODES_MAX, f);
    fclose(f);

    int result = dspRuntimeInit(opcodes, OPCODES_MAX, fs, 12345, 24);          /* avdsp_plugin.c:316 */
    if (result < 0) { fprintf(stderr, "dspRuntimeInit: %d (%s)\n", result, dspRuntimeLastError()); return 1; }
    int *dataPtr = (int *)opcodes + result;                                   /* avdsp_plugin.c:322 */
    (void)words;

    opcode_t *codestart[CORES_MAX];
    int nbcores = 0;
    for (; nbcores < CORES_MAX; nbcores++) {
        opcode_t *core = dspFindCore(opcodes, nbcores + 1);
        if (!core) break;
        codestart[nbcores] = dspFindCoreBegin(core);
    }

    f = fopen(argv[3], "rb");
    if (!f) { perror(argv[3]); return 1; }
    fseek(f, 0, SEEK_END);
    long frames = ftell(f) / (4L * nbchin);
    fseek(f, 0, SEEK_SET);
    int *src = (int *)malloc((size_t)frames * nbchin * 4), *dst = (int *)calloc((size_t)frames * nbchout, 4);
    if (fread(src, 4, (size_t)frames * nbchin, f) != (size_t)frames * nbchin) { fprintf(stderr, "short read\n"); return 1; }
    fclose(f);

    for (long n0 = 0; n0 < frames; n0 += block) {
        int size = (int)(frames - n0 < block ? frames - n0 : block);
        for (int nc = 0; nc < nbcores; nc++) {                                 /* avdsp_plugin.c:95-142: cores outer, frames inner */
            int rc = DSP_RUNTIME_FORMAT(dspRuntimeBlock)(codestart[nc], dataPtr,
                         (const void *)(src + n0 * nbchin), nbchin, in_base,
                         (void *)(dst + n0 * nbchout), nbchout, out_base, size);
            if (rc < 0) { fprintf(stderr, "core %d: %d (%s)\n", nc + 1, rc, dspRuntimeLastError()); return 1; }
        }
    }
    if (dspRuntimeSyncState(dataPtr) < 0) { fprintf(stderr, "sync: %s\n", dspRuntimeLastError()); return 1; }

    f = fopen(argv[6], "wb");
    fwrite(dst, 4, (size_t)frames * nbchout, f);
    fclose(f);
    printf("cores=%d frames=%ld state[0..3]=%08x %08x %08x %08x\n", nbcores, frames,
           (unsigned)dataPtr[0], (unsigned)dataPtr[1], (unsigned)dataPtr[2], (unsigned)dataPtr[3]);
    dspRuntimeRelease();
    return 0;
}
