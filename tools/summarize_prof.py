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
    m = re.search(r"(fir_stream|fir_tile|fir_mfma|fir_plain|biquad_row_i64|biquad_row|biquad_pipe|biquad_simple|strand_lanes|tpdf_walk|interp_wave_instances|interp_wave_grid|interp_wave|interp_core|chain_lane|chain_rows|fir_lane_history|fir_lane_state|fir_lane_feed|fir_lane_hw|fir_lane|interp_wave_instances|passthrough)(<[^>]*>)?", name)
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


def traffic(d, workload, out_json):
    """FETCH_SIZE/WRITE_SIZE (KiB) per dispatch -> HBM bytes per launch per kernel, merged into out_json.
    FETCH_SIZE is doubled: on gfx950 it reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM)."""
    import json
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(d, "pmc_*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                agg[k.split("<")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    try:
        data = json.load(open(out_json))
    except (OSError, ValueError):
        data = {}
    w = data.setdefault(workload, {})
    for k, cs in agg.items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            fetch = sum(cs["FETCH_SIZE"]) / len(cs["FETCH_SIZE"]) * 1024 * 2
            write = sum(cs["WRITE_SIZE"]) / len(cs["WRITE_SIZE"]) * 1024
            w[k] = round(fetch + write)
    json.dump(data, open(out_json, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    if len(sys.argv) >= 5 and sys.argv[2] == "--traffic":
        traffic(sys.argv[1], sys.argv[3], sys.argv[4])
    else:
        main(sys.argv[1])
