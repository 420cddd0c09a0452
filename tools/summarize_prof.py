#!/usr/bin/env python3
"""Condense a tools/profile_gpu.sh output directory (rocprofv3 csv files) into one markdown summary.
usage: tools/summarize_prof.py gpurun_out/prof_TAG > profiles/TAG_summary.md"""
import collections
import csv
import glob
import os
import re
import sys


def short(name):
    m = re.search(r"(fir_mfma|fir_plain|biquad_pipe|biquad_simple|passthrough)<[^>]*>", name)
    return m.group(0) if m else None


def main(d):
    print(f"# rocprofv3 summary: {os.path.basename(d.rstrip('/'))}\n")
    for f in glob.glob(os.path.join(d, "trace", "*", "*_kernel_stats.csv")):
        print("## kernel trace (`rocprofv3 --kernel-trace --stats`)\n")
        print("| kernel | calls | avg us | min us | max us | % |")
        print("|---|---|---|---|---|---|")
        for r in csv.DictReader(open(f)):
            k = short(r["Name"]) or r["Name"][:60]
            print(f"| `{k}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | "
                  f"{float(r['MaxNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
        print()
    log = os.path.join(d, "trace.log")
    if os.path.exists(log):
        for line in open(log):
            if line.startswith("{"):
                print("bench line of the traced run:\n\n```\n" + line.strip() + "\n```\n")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for f in glob.glob(os.path.join(d, "pmc_*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not k:
                continue
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"])
    if agg:
        print("## PMC counters (separate `rocprofv3 --pmc` passes, mean per dispatch)\n")
    for k, cs in agg.items():
        g = meta[k]
        print(f"### `{k}`  grid {g[0]} wg {g[1]} LDS {g[2]} B VGPR {g[3]} AGPR {g[4]} SGPR {g[5]} scratch {g[6]}\n")
        print("| counter | mean per dispatch | dispatches |")
        print("|---|---|---|")
        for c in sorted(cs):
            v = cs[c]
            print(f"| {c} | {sum(v)/len(v):.6g} | {len(v)} |")
        m = {c: sum(v) / len(v) for c, v in cs.items()}
        notes = []
        if "FETCH_SIZE" in m:
            notes.append(f"FETCH_SIZE {m['FETCH_SIZE']:.0f} KiB raw; x2 per the gfx950 correction for wide coalesced reads = {2*m['FETCH_SIZE']*1024/1e6:.1f} MB (uncorrected {m['FETCH_SIZE']*1024/1e6:.1f} MB)")
        if "WRITE_SIZE" in m:
            notes.append(f"WRITE_SIZE {m['WRITE_SIZE']:.0f} KiB = {m['WRITE_SIZE']*1024/1e6:.1f} MB")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CYCLES" in m and m["SQ_BUSY_CYCLES"]:
            notes.append(f"SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES = {m['SQ_VALU_MFMA_BUSY_CYCLES']/m['SQ_BUSY_CYCLES']:.3f}")
        if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE"):
            notes.append(f"LDS bank-conflict cycles / LDS active cycles = {m['SQ_LDS_BANK_CONFLICT']/m['SQ_LDS_IDX_ACTIVE']:.3f}")
        if "SQ_WAVE_CYCLES" in m and "SQ_WAIT_ANY" in m:
            pass
        for n in notes:
            print(f"\n* {n}")
        print()


if __name__ == "__main__":
    main(sys.argv[1])
