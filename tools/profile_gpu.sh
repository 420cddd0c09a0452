#!/usr/bin/env bash
# Run on the GPU box (via gpurun) from the repo root: kernel trace + PMC passes of bench.py.
# usage: tools/profile_gpu.sh TAG [bench.py args...]      -> gpurun_out/prof_TAG/
set -uo pipefail
TAG="$1"; shift
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS=(--no-cpu-baseline --steps 20 --warmup 5 "$@")
# The counter passes run the kernels ONE AT A TIME: rocprofv3 --pmc serialises dispatches, and a launch arrangement that needs two
# queues to run at once (the overlap mode, its ready words) is then not the thing measured -- round 4's north PMC table counted
# seconds of FIR waves polling for a cascade that could not start (SQ_WAVE_CYCLES 1.2e12 for a 0.5-ms kernel).  The library now
# sees that by itself (side_by_side 0: events instead of ready words), but the passes say so explicitly: back to back, no polling.
# The kernel-trace pass above them is the timed arrangement as it ships.
PMC_ARGS=("${ARGS[@]}" --overlap 0 --ready-words 0 --no-verify)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" "${ARGS[@]}" > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
# counters in their own runs (no tracing domains mixed in), split by hardware block budget
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -- python3 "$ROOT/bench.py" "${PMC_ARGS[@]}" > "$OUT/pmc_sq.log" 2>&1
echo "pmc_sq rc=$?"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/pmc_sq2" -- python3 "$ROOT/bench.py" "${PMC_ARGS[@]}" > "$OUT/pmc_sq2.log" 2>&1
echo "pmc_sq2 rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" "${PMC_ARGS[@]}" > "$OUT/pmc_fetch.log" 2>&1
echo "pmc_fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" "${PMC_ARGS[@]}" > "$OUT/pmc_write.log" 2>&1
echo "pmc_write rc=$?"
# condense on the box (the per-dispatch csv files of a 0.5 s settle phase run to tens of MB): summary, kernel stats, traffic
cd "$ROOT"
python3 tools/summarize_prof.py "$OUT" > "gpurun_out/${TAG}_summary.md" || echo "summarize_prof: a counter table was REFUSED (see the summary)"
cp "$OUT"/trace/*/*_kernel_stats.csv "gpurun_out/${TAG}_kernel_stats.csv"
[ -n "${TRAFFIC_KEY:-}" ] && python3 tools/summarize_prof.py "$OUT" --traffic "$TRAFFIC_KEY" gpurun_out/traffic.json
rm -rf "$OUT"
