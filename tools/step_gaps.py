#!/usr/bin/env python3
"""Idle time between the kernels of a step, from a rocprofv3 kernel trace of bench.py (the csv's start/end timestamps):
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 $REPO/bench.py --shard 0/8 --overlap 0 --no-cpu-baseline
    python tools/step_gaps.py /tmp/tr"""
import csv
import glob
import sys

import numpy as np

rows = []
for f in glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[len(rows) // 2:]                      # the second half: clock settled, timed region
short = lambda n: "fir" if "fir_" in n else "biquad" if "biquad_" in n else "other"
gaps = {"biquad->fir": [], "fir->biquad": []}
dur = {"fir": [], "biquad": []}
for (s0, e0, n0), (s1, e1, n1) in zip(rows[:-1], rows[1:]):
    a, b = short(n0), short(n1)
    if a in dur:
        dur[a].append(e0 - s0)
    if f"{a}->{b}" in gaps:
        gaps[f"{a}->{b}"].append(s1 - e0)
for k, v in dur.items():
    if v:
        print(f"{k}: median {np.median(v) / 1e3:.1f} us ({len(v)} launches)")
for k, v in gaps.items():
    if v:
        print(f"gap {k}: median {np.median(v) / 1e3:.1f} us, p10 {np.percentile(v, 10) / 1e3:.1f}, p90 {np.percentile(v, 90) / 1e3:.1f}")
# the overlap mode: the FIRs follow each other on the caller's stream, the cascades run beside them
firs = [(s_, e_) for s_, e_, n_ in rows if short(n_) == "fir"]
if len(firs) > 8:
    ff = [b[0] - a[1] for a, b in zip(firs[:-1], firs[1:])]
    st = [b[0] - a[0] for a, b in zip(firs[:-1], firs[1:])]
    print(f"fir end -> next fir start: median {np.median(ff) / 1e3:.1f} us (p10 {np.percentile(ff, 10) / 1e3:.1f}, p90 {np.percentile(ff, 90) / 1e3:.1f}); fir start -> next fir start {np.median(st) / 1e3:.1f} us")
# a few consecutive launches in full, relative times
t0 = rows[0][0]
print("start_us  end_us  dur_us  kernel")
for s_, e_, n_ in rows[40:52]:
    print(f"{(s_ - t0) / 1e3:9.1f} {(e_ - t0) / 1e3:9.1f} {(e_ - s_) / 1e3:7.1f}  {short(n_)}")
