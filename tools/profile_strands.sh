#!/usr/bin/env bash
# Run on the GPU box (via gpurun) from the repo root: kernel trace of the 4096-strand crossover core (tools/wide_core_bench.py).
# usage: tools/profile_strands.sh TAG      -> gpurun_out/TAG_summary.md, gpurun_out/TAG_kernel_stats.csv
set -uo pipefail
TAG="$1"
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/tools/wide_core_bench.py" --strands 4096 --lanes-only > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
cd "$ROOT"
python3 tools/summarize_prof.py "$OUT" > "gpurun_out/${TAG}_summary.md"
grep "strands in one core" "$OUT/trace.log" | sed 's/^/    /' >> "gpurun_out/${TAG}_summary.md"
cp "$OUT"/trace/*/*_kernel_stats.csv "gpurun_out/${TAG}_kernel_stats.csv"
rm -rf "$OUT"
