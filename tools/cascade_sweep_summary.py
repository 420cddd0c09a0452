#!/usr/bin/env python3
"""tools/cascade_sweep.sh's table: per workload the bench line's launch time and roofs, and the SQ counters of the cascade kernel
(mean per dispatch of the separate --pmc pass)."""
import collections
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
rows = []
for w in ("cfg3", "cfg3x4", "cfg3x8", "cfg3i", "cfg3ix4", "cfg3ix8"):
    line = None
    try:
        for l in open(os.path.join(d, w + ".json")):
            if l.startswith("{"):
                line = json.loads(l)
    except OSError:
        pass
    if not line:
        continue
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "pmc_" + w, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "biquad_row" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    m = {k: sum(v) / len(v) for k, v in agg.items()}
    rows.append((w, line, m))
print("| workload | chains | waves per SIMD | launch us | step us | Gsamples/s | frac of the VALU roof | HBM frac | VALU insts / (wave x step) | SQ_WAIT_ANY / SQ_WAVE_CYCLES | SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES | busy cycles per step and wave |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for w, line, m in rows:
    C = line["config"]["channels"]; B = line["config"]["block"]; S = line["config"]["sections"]
    r = line["roofline"]
    waves = m.get("SQ_WAVES", C / 4.0)
    steps = B + 2 * (S - 1) + 1
    ipw = m["SQ_INSTS_VALU"] / waves / steps if "SQ_INSTS_VALU" in m else float("nan")
    wait = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else float("nan")
    act = m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") and "SQ_ACTIVE_INST_VALU" in m else float("nan")
    cyc = 4.0 * m["SQ_WAVE_CYCLES"] / waves / steps if m.get("SQ_WAVE_CYCLES") else float("nan")      # SQ_WAVE_CYCLES counts quad-cycles
    print(f"| {w} | {C} | {C / 4 / 1024:.0f} | {r['launch_ms'] * 1e3:.1f} | {line['ms_per_step'] * 1e3:.1f} | {line['value'] / 1e3:.1f} | {r['frac']:.3f} | {r['hbm_frac']:.3f} | "
          f"{ipw:.1f} | {wait:.2f} | {act:.2f} | {cyc:.0f} |")
print()
print("`frac of the VALU roof`: 10 flop per section and sample against 78.6 TFLOP/s (format 6), 5 MADs against a quarter of the FP32 FMA rate (int64).  "
      "`busy cycles per step and wave`: 4 x SQ_WAVE_CYCLES / waves / (1024 + 31) -- what a wave's step costs in wall cycles while it shares its SIMD.")
