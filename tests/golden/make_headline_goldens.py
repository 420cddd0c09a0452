#!/usr/bin/env python3
"""Full-width pins of the headline configurations (BASELINE.json configs 3, 4, 5 and the north-star
program) from the COMPILED REFERENCE -- every channel, several blocks, outputs and final state.

Runs only in the build container (needs /root/reference and oracle/_ref built by oracle/build_ref.sh):

    python tests/golden/make_headline_goldens.py [case ...]

The reference runtime is single-threaded and keeps one program per process, so the channels of a case are
cut into groups; each group is the same filters, impulses and input columns as in the unsharded program
(progbuilder.synth_program(channel_base=...)), run by oracle/_ref/ref_driver in its own process exactly as
the reference hosts run it (blocks of `block` frames, cores outer, frames inner).  Channels of these
programs are independent chains, so the unsharded result is the concatenation of the groups' -- which the
script first PROVES on the reference itself at a small size (unsharded run == concatenated group runs,
outputs and state) before it trusts the cut.

What is committed per case (tests/golden/headline_<case>.npz + headline_manifest.json), a few tens of KB:
    out_sha      SHA-256 of the whole [frames][C] output            block_sha  one per block of frames
    state_sha    SHA-256 of the final state area (unsharded layout)
    col_sum      per channel: sum of the raw 32-bit output words over all frames (mod 2^32)
    row_sum      per frame: sum over all channels (mod 2^32)         -> a mismatch is localised to (channel, frame)
    state_col    per channel: sum of its state words (mod 2^32)
    head / tail  the first / last 2 frames, every channel
The fixtures are data: seeds, shapes, hashes and sums; no reference source.
"""
from __future__ import annotations

import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from avdsp_amd import progbuilder as pb          # noqa: E402
from oracle import pyoracle as po                # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
WORKERS = int(os.environ.get("AVDSP_GOLDEN_WORKERS", "7"))

# name: (fmt, channels, sections, taps, frames, block, channels per reference process)
CASES = {
    "cfg3_f2":  (2, 4096, 16, 0, 2048, 1024, 256),
    "cfg3_f4":  (4, 4096, 16, 0, 2048, 1024, 256),
    "cfg3_f6":  (6, 4096, 16, 0, 2048, 1024, 256),
    "cfg4_f4":  (4, 256, 0, 4096, 5120, 1024, 8),
    "cfg4_f6":  (6, 256, 0, 4096, 5120, 1024, 8),
    "cfg5_f6":  (6, 16384, 8, 2048, 3072, 1024, 64),
    "north_f6": (6, 4096, 16, 4096, 5120, 1024, 32),
}
SEED = 20260104


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def words(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a).view(np.uint32)


def run_group(fmt, S, T, lo, hi, x_cols, block, tmp):
    """One reference process over channels [lo, hi): returns (out [frames][hi-lo], state words)."""
    n = hi - lo
    prog = pb.synth_program(fmt, n, S, T, channel_base=lo)
    rc, out, buf = po.run_reference(fmt, prog, x_cols, n, n, 0, block=block, tmpdir=tmp, want_state=True)
    assert rc >= 0, (lo, hi, rc)
    return out, buf[rc:rc + int(prog[2])]


def run_case_groups(fmt, C, S, T, frames, block, group, tmp, progress=None):
    fl = fmt in (5, 6)
    x = pb.lcg_input(frames, C, fl, seed=SEED)
    spans = [(lo, min(lo + group, C)) for lo in range(0, C, group)]
    t0 = time.time()
    done = [0]

    def job(span):
        lo, hi = span
        r = run_group(fmt, S, T, lo, hi, np.ascontiguousarray(x[:, lo:hi]), block, tmp)
        done[0] += 1
        if progress and done[0] % 8 == 0:
            print(f"    {progress}: {done[0]}/{len(spans)} groups, {time.time() - t0:.0f} s", flush=True)
        return r

    with ThreadPoolExecutor(WORKERS) as ex:
        res = list(ex.map(job, spans))
    out = np.concatenate([r[0] for r in res], axis=1)
    state = np.concatenate([r[1] for r in res])
    return x, out, state, spans


def prove_cut_on_the_reference(tmp):
    """unsharded reference run == concatenation of the groups' runs (outputs and state), small sizes"""
    for fmt, C, S, T, frames, block, group in ((6, 12, 3, 40, 300, 128, 4), (2, 10, 4, 0, 200, 64, 3), (4, 6, 2, 18, 150, 50, 4)):
        x, out, state, _ = run_case_groups(fmt, C, S, T, frames, block, group, tmp)
        prog = pb.synth_program(fmt, C, S, T)
        rc, ref_out, buf = po.run_reference(fmt, prog, x, C, C, 0, block=block, tmpdir=tmp, want_state=True)
        ref_state = buf[rc:rc + int(prog[2])]
        assert (words(out) == words(ref_out)).all(), ("outputs differ from the unsharded reference run", fmt)
        assert len(state) == len(ref_state) and (state == ref_state).all(), ("state differs from the unsharded reference run", fmt)
    print("  cut proven on the reference: unsharded == concatenated groups (outputs and state), formats 6, 2, 4")


def main():
    if not po.have_ref():
        sys.exit("oracle/_ref is missing: run oracle/build_ref.sh in the build container first")
    names = sys.argv[1:] or list(CASES)
    mpath = os.path.join(OUT, "headline_manifest.json")
    manifest = {}
    if os.path.exists(mpath):
        with open(mpath) as f:
            manifest = json.load(f)["cases"]
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        prove_cut_on_the_reference(tmp)
        for name in names:
            fmt, C, S, T, frames, block, group = CASES[name]
            t0 = time.time()
            x, out, state, spans = run_case_groups(fmt, C, S, T, frames, block, group, tmp, progress=name)
            per_ch = len(state) // C
            assert per_ch * C == len(state)
            w = words(out)
            entry = dict(fmt=fmt, channels=C, sections=S, taps=T, frames=frames, block=block, seed=SEED,
                         group=group, in_sha=sha(x), out_sha=sha(out), state_sha=sha(state),
                         block_sha=[sha(out[b:b + block]) for b in range(0, frames, block)],
                         state_words_per_channel=per_ch)
            np.savez_compressed(os.path.join(OUT, f"headline_{name}.npz"),
                                col_sum=w.sum(axis=0, dtype=np.uint64).astype(np.uint32),
                                row_sum=w.sum(axis=1, dtype=np.uint64).astype(np.uint32),
                                state_col=state.reshape(C, per_ch).sum(axis=1, dtype=np.uint64).astype(np.uint32),
                                head=out[:2], tail=out[-2:])
            manifest[name] = entry
            print(f"  {name}: {C} ch x {frames} frames through the reference in {time.time() - t0:.0f} s  out_sha {entry['out_sha'][:16]}", flush=True)
            with open(mpath, "w") as f:
                json.dump(dict(cases=manifest,
                               note="generated by tests/golden/make_headline_goldens.py from the compiled reference (oracle/_ref); "
                                    "program = progbuilder.synth_program(fmt, channels, sections, taps), input = "
                                    "progbuilder.lcg_input(frames, channels, fmt in (5, 6), seed)"), f, indent=1)


if __name__ == "__main__":
    main()
