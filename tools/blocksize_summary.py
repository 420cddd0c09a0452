#!/usr/bin/env python3
"""tools/blocksize_sweep.sh's bench lines -> a markdown table: step, ns per frame of the block, kernel times, fraction of the roof."""
import json
import sys


def main(path):
    rows = [json.loads(l) for l in open(path) if l.startswith("{")]
    print("| workload | block (frames) | step (us) | ns per frame | Gsamples/s | dominant kernel: launch (us) | frac of its roof | cascade alone (us) | verified |")
    print("|---|---|---|---|---|---|---|---|---|")
    for l in rows:
        if "failed" in l:
            print(f"| {l['failed']} | FAILED |")
            continue
        c, r = l["config"], l["roofline"] or {}
        B = c["block"]
        us = l["ms_per_step"] * 1e3
        name = c["workload"].split(":")[0]
        print(f"| {name} | {B} | {us:.1f} | {us * 1e3 / B:.1f} | {l['value'] / 1e3:.2f} | {r.get('kernel', '-')} {r.get('launch_ms', 0) * 1e3:.1f} "
              f"({r.get('launches', 0)} stamps) | {r.get('frac', 0):.3f} ({r.get('bound', '-')}) | {l['kernels_ms']['biquad'] * 1e3:.1f} | {'yes' if l.get('verified') else 'no'} |")


if __name__ == "__main__":
    main(sys.argv[1])
