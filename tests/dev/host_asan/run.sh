#!/bin/bash
# The C host runtime under AddressSanitizer, driven by the -m gpu tests, on a machine WITHOUT a GPU:  bash tests/dev/host_asan/run.sh [pytest args]
# (stub_device.c stands in for the kernels and computes nothing: almost every test FAILS on its sample comparison.  What counts is
# the last line: sanitizer reports found in the log, or none.)
set -u
ROOT="$(cd "$(dirname "$0")/../../.." && pwd)"
OUT="${HOST_ASAN_OUT:-/tmp/host_asan}"
mkdir -p "$OUT"
gcc -fsanitize=address -fno-omit-frame-pointer -g -O1 -std=gnu99 -fPIC -shared -I"$ROOT/include" -o "$OUT/libavdsp_stub.so" \
    "$ROOT/avdsp_amd/csrc/avdsp_host.c" "$ROOT/avdsp_amd/csrc/avdsp_qformat.c" "$ROOT/tests/dev/host_asan/stub_device.c" -lm || exit 2
cd "$ROOT"
AVDSP_LIB="$OUT/libavdsp_stub.so" PYTHONPATH="$ROOT/tests/dev/host_asan:${PYTHONPATH:-}" \
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so)" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
    python -m pytest "${@:-tests}" -q -m gpu -p no:cacheprovider -p fake_cuda > "$OUT/run.log" 2>&1
tail -1 "$OUT/run.log"
if grep -q "AddressSanitizer" "$OUT/run.log"; then grep -n -A 30 "AddressSanitizer" "$OUT/run.log" | head -80; echo "host_asan: REPORTS (see $OUT/run.log)"; exit 1; fi
echo "host_asan: no sanitizer report"
