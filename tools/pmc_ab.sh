#!/usr/bin/env bash
# A/B of two library builds under the SQ counters that matter for the FIR loop.  usage: tools/pmc_ab.sh TAG [LIB]
set -uo pipefail
TAG="$1"; LIB="${2:-}"
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/ab_$TAG"; mkdir -p "$OUT"
[ -n "$LIB" ] && export AVDSP_LIB="$LIB"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/pmc" -- python3 "$ROOT/bench.py" --no-cpu-baseline --steps 10 --warmup 2 > "$OUT/pmc.log" 2>&1
echo "rc=$?"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_BUSY_CYCLES": n[k] += 1
for k in tot:
    if "fir_mfma" in k or "biquad" in k:
        print(k, n[k], {c: round(v / max(n[k], 1)) for c, v in tot[k].items()})
PY
