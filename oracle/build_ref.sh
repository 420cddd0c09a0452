#!/usr/bin/env bash
# build_ref.sh -- TEST INFRASTRUCTURE.  Compiles the *reference* AVDSP sources where they lie under
# /root/reference into oracle/_ref/ (git-ignored binaries only; no reference source is copied).
# Only runs where /root/reference exists (the build container); the GPU box uses the prebuilt files.
#
# What gets built, with the reference's own flags (runtime/Makefile:13,18,41-45:
# gcc -DLINUX -std=gnu99 -Ofast -fPIC -shared -DDSP_FORMAT=N):
#
#   libavdspref_{2,3,4,5,6}.so  the complete reference runtime, one per DSP_FORMAT.
#       dsp_runtime.c does not compile as shipped: four lines, none on the hot
#       path, are repaired ON THE FLY by the sed expressions below while the file is streamed to
#       gcc; nothing patched is written to disk:
#         :208   dspQNM(p->f,new)            -> dspQNM(p->f,32-new,new)   (prototype takes 3 args)
#         :1288  bad cast in DSP_SINE        -> ((dspParam_t*)cptr)[dspSamplingFreqIndex]
#         :1291  missing ';'
#         :1304  stray #endif
#       The int build keeps two undefined symbols (dspQNMmax, DSP_Q31: called by DIRAC/SQUAREWAVE/
#       SINE only, never defined anywhere in the reference).  No stand-ins are written for them:
#       the libraries are opened with RTLD_LAZY by ref_driver, so they are simply never bound.
#   refk_{2,6}.so               oracle/ref_kernels.c, which #includes the UNPATCHED reference headers
#       (dsp_biquadSTD.h, dsp_firSTD.h, dsp_ieee754.h, dsp_fpmath.h) and exports the hot kernels.
#   libavdspencoder.so, ref_encode, ref_encode_ops   the unmodified reference encoder + the oracle/ref_encode*.c drivers.
#   ref_driver                  oracle/ref_driver.c: runs a .bin over a raw input file through a
#                               libavdspref_N.so and writes the raw output (used to make goldens).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${AVDSP_REFERENCE:-/root/reference}/module_avdsp"
OUT="$HERE/_ref"
if [ ! -d "$REF/runtime" ]; then
    echo "build_ref.sh: $REF not present; keeping whatever is in $OUT" >&2
    exit 0
fi
mkdir -p "$OUT"
RT="$REF/runtime"
ENC="$REF/encoder"
CF="-DLINUX -std=gnu99 -Ofast -fPIC -w"

for F in 2 3 4 5 6; do
    sed -e '208s/dspQNM(p->f, new)/dspQNM(p->f, 32-new, new)/' \
        -e '1288s/.*/            dspParam_t epsilon = ((dspParam_t*)cptr)[dspSamplingFreqIndex];/' \
        -e '1291s/(\*gainPtr)) \/\/force/(*gainPtr)); \/\/force/' \
        -e '1304d' "$RT/dsp_runtime.c" |
    gcc $CF -I"$RT" -DDSP_FORMAT=$F -shared -o "$OUT/libavdspref_$F.so" -x c - "$RT/dsp_header.c" -lm
done

for F in 2 6; do
    gcc $CF -I"$RT" -DDSP_FORMAT=$F -shared -o "$OUT/refk_$F.so" "$HERE/ref_kernels.c" -lm
done

# the encoder builds unmodified (encoder/Makefile:18-20,56-60); DSP_PRINTF left off to keep it quiet
gcc $CF -I"$ENC" -I"$RT" -shared -o "$OUT/libavdspencoder.so" \
    "$ENC/dsp_encoder.c" "$ENC/dsp_fileaccess.c" "$ENC/dsp_filters.c" \
    "$ENC/dsp_HilbertDesign.c" "$ENC/dsp_nanosharcxml.c" "$RT/dsp_header.c" -lm
gcc $CF -I"$ENC" -I"$RT" -o "$OUT/ref_encode" "$HERE/ref_encode.c" \
    -L"$OUT" -lavdspencoder -Wl,-rpath,'$ORIGIN' -lm -ldl
gcc $CF -I"$ENC" -I"$RT" -o "$OUT/ref_encode_ops" "$HERE/ref_encode_ops.c" \
    -L"$OUT" -lavdspencoder -Wl,-rpath,'$ORIGIN' -lm -ldl
gcc $CF -I"$ENC" -I"$RT" -o "$OUT/enc_sweep" "$HERE/enc_sweep.c" \
    -L"$OUT" -lavdspencoder -Wl,-rpath,'$ORIGIN' -lm -ldl
gcc $CF -I"$ENC" -I"$RT" -o "$OUT/enc_hilbert" "$HERE/enc_hilbert.c" \
    -L"$OUT" -lavdspencoder -Wl,-rpath,'$ORIGIN' -lm -ldl
gcc $CF -I"$ENC" -I"$RT" -o "$OUT/dspcreate" "$ENC/dspcreate.c" \
    -L"$OUT" -lavdspencoder -Wl,-rpath,'$ORIGIN' -lm -ldl
for P in crossoverLV6 oktodac_diy testfunction; do
    gcc $CF -I"$ENC" -I"$RT" -shared -o "$OUT/$P.so" "$REF/dspprogs/$P.c" \
        -L"$OUT" -lavdspencoder -Wl,-rpath,'$ORIGIN' -lm
done

gcc -O2 -std=gnu99 -w -I"$HERE/../include" -o "$OUT/ref_driver" "$HERE/ref_driver.c" -ldl
echo "build_ref.sh: reference binaries in $OUT"
