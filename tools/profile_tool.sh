#!/usr/bin/env bash
# Run on the GPU box (via gpurun) from the repo root: rocprofv3 kernel trace of one of the side benches under tools/.
# usage: tools/profile_tool.sh TAG tools/SCRIPT.py [args...]   -> gpurun_out/TAG_summary.md (kernel table + the script's own lines),
#        gpurun_out/TAG_kernel_stats.csv.      e.g.  tools/profile_tool.sh r03_strands4096 tools/wide_core_bench.py --strands 4096 --lanes-only
set -uo pipefail
TAG="$1"; shift
SCRIPT="$1"; shift
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/$SCRIPT" "$@" > "$OUT/trace.log" 2>&1
echo "trace rc=$?"
cd "$ROOT"
python3 tools/summarize_prof.py "$OUT" > "gpurun_out/${TAG}_summary.md"
echo "output of \`$SCRIPT $*\` under the profiler:" >> "gpurun_out/${TAG}_summary.md"
echo >> "gpurun_out/${TAG}_summary.md"
grep -v "amdgpu.ids\|^\[\|rocprofv3\|^W2\|^E2" "$OUT/trace.log" | sed 's/^/    /' >> "gpurun_out/${TAG}_summary.md"
cp "$OUT"/trace/*/*_kernel_stats.csv "gpurun_out/${TAG}_kernel_stats.csv"
rm -rf "$OUT"
