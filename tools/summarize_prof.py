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


CLOCK_GHZ = 2.4                          # MI355X peak engine clock (MI355X_MICROARCH.md)


def main(d):
    """returns the number of counter tables refused"""
    print(f"# rocprofv3 summary: {os.path.basename(d.rstrip('/'))}\n")
    print("Kernel trace: the arrangement as it ships (overlap mode, the library's choice of ready words).  Counter passes: "
          "`--overlap 0 --ready-words 0` -- `rocprofv3 --pmc` serialises dispatches, so the kernels are measured one at a time, back to back, "
          "and no wave polls for another queue's kernel (tools/profile_gpu.sh).\n")
    avg_ns = {}
    for f in glob.glob(os.path.join(d, "trace", "*", "*_kernel_stats.csv")):
        print("## kernel trace (`rocprofv3 --kernel-trace --stats`)\n")
        print("| kernel | calls | avg us | min us | max us | % |")
        print("|---|---|---|---|---|---|")
        for r in csv.DictReader(open(f)):
            k = short(r["Name"]) or r["Name"][:60]
            avg_ns.setdefault(k, float(r["AverageNs"]))
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
    refused = 0
    for k, cs in agg.items():
        g = meta[k]
        # A table is only the kernel's if its waves ran for about as long as the kernel does: SQ_WAVE_CYCLES (summed over waves; the
        # counter ticks once per 4 clocks per wave on gfx950: round 3's valid tables read 1.9e7 for 1024 waves x 35 us) must not exceed
        # waves x launch time x clock x 1.5.  Round 4's north table did, 60-fold: polling waves under a serialising profiler.
        mw = {c: sum(v) / len(v) for c, v in cs.items()}
        t_ns = avg_ns.get(k)
        if t_ns and mw.get("SQ_WAVES") and mw.get("SQ_WAVE_CYCLES"):
            bound = mw["SQ_WAVES"] * t_ns * CLOCK_GHZ * 1.5
            if mw["SQ_WAVE_CYCLES"] > bound:
                refused += 1
                print(f"### `{k}`: counter table REFUSED\n\nSQ_WAVE_CYCLES {mw['SQ_WAVE_CYCLES']:.3g} exceeds waves x launch time x {CLOCK_GHZ} GHz x 1.5 = "
                      f"{bound:.3g} ({mw['SQ_WAVES']:.0f} waves, {t_ns / 1e3:.1f} us in the kernel trace): the waves of these passes did something "
                      f"else than the kernel does when timed (polling for another queue's kernel under the profiler's serialised dispatches?).\n")
                continue
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
        if m.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in m:
            notes.append(f"SQ_WAIT_ANY / SQ_WAVE_CYCLES = {m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES']:.3f}")
        if m.get("SQ_INSTS_MFMA") and "SQ_INSTS_VALU" in m:
            notes.append(f"non-MFMA VALU instructions per MFMA = {(m['SQ_INSTS_VALU'] - m['SQ_INSTS_MFMA'])/m['SQ_INSTS_MFMA']:.2f} "
                         f"(SQ_INSTS_VALU counts the MFMAs too); LDS instructions per MFMA = {m.get('SQ_INSTS_LDS', 0)/m['SQ_INSTS_MFMA']:.2f}")
        if t_ns and m.get("SQ_WAVES") and m.get("SQ_WAVE_CYCLES"):
            notes.append(f"SQ_WAVE_CYCLES / (waves x launch time x {CLOCK_GHZ} GHz) = {m['SQ_WAVE_CYCLES']/(m['SQ_WAVES']*t_ns*CLOCK_GHZ):.3f} "
                         f"(launch time {t_ns/1e3:.1f} us from the kernel trace)")
        for n in notes:
            print(f"\n* {n}")
        print()
    return refused


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
        sys.exit(1 if main(sys.argv[1]) else 0)
