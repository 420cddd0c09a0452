#!/usr/bin/env bash
# The cascade kernels from one wave per SIMD (4096 chains x 16 sections) to eight (32768, the occupancy limit): launch time, instructions and waits.
# Run on the GPU box (via gpurun) from the repo root:  bash tools/cascade_sweep.sh TAG   -> gpurun_out/TAG_cascade_sweep.md
set -uo pipefail
TAG="${1:-r04}"
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/prof_${TAG}_sweep"
mkdir -p "$OUT"
MD="$ROOT/gpurun_out/${TAG}_cascade_sweep.md"
echo "# biquad_row / biquad_row_i64 at 1, 4 and 8 waves per SIMD (16 sections, blocks of 1024 frames; bench.py --workload ...)" > "$MD"
echo >> "$MD"
cd /tmp && export TMPDIR=/tmp
for W in cfg3 cfg3x4 cfg3x8 cfg3i cfg3ix4 cfg3ix8; do
  python3 "$ROOT/bench.py" --workload $W --no-cpu-baseline --steps 32 --warmup 3 > "$OUT/$W.json" 2> "$OUT/$W.err"
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/pmc_$W" -- python3 "$ROOT/bench.py" --workload $W --no-cpu-baseline --steps 8 --warmup 2 --settle 0.1 > "$OUT/pmc_$W.log" 2>&1
  echo "$W rc=$?"
done
cd "$ROOT"
python3 tools/cascade_sweep_summary.py "$OUT" >> "$MD"
rm -rf "$OUT"
cat "$MD"
