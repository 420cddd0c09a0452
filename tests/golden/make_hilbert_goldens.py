#!/usr/bin/env python3
"""Fixtures for dsp_Hilbert() (round 5; f1 completeness), from the COMPILED REFERENCE -- build container only.

  enc_hilbert_<fmt>_<fmin>_<fmax>.bin   what the reference ENCODER (oracle/_ref/libavdspencoder.so) makes of oracle/enc_hilbert.c
                                        (both branches, 1 .. 10 stages, ten transition widths, four encodings / rate ranges)
  hilbert_f<N>.npz + hilbert_manifest.json
                                        those programs through the reference RUNTIME (oracle/_ref/libavdspref_N.so via ref_driver):
                                        22 output channels, 600 frames of the seeded LCG input, output and final state

The fixtures are data (program words, outputs); no reference source is copied.  Kept apart from make_goldens.py so that adding
them does not rewrite the 112 older fixtures."""
from __future__ import annotations

import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po                # noqa: E402
from tests.golden.make_goldens import run_case, lcg    # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
VARIANTS = [(2, 4, 9), (6, 4, 7), (6, 0, 13), (2, 5, 5)]


def main():
    if not po.have_ref() or not os.path.exists(os.path.join(po.REF_DIR, "enc_hilbert")):
        sys.exit("oracle/_ref/enc_hilbert is missing: run oracle/build_ref.sh in the build container first")
    for fmt, fmin, fmax in VARIANTS:
        subprocess.check_call([os.path.join(po.REF_DIR, "enc_hilbert"), str(fmt), str(fmin), str(fmax),
                               os.path.join(OUT, f"enc_hilbert_{fmt}_{fmin}_{fmax}.bin")], stdout=subprocess.DEVNULL)
    manifest = []
    # the Q28 encoding runs in the int64 model, the float encoding in the four others; 48 kHz and (wide range) 96 kHz
    for fmt, fname, fs in ((2, "enc_hilbert_2_4_9.bin", 48000), (2, "enc_hilbert_2_4_9.bin", 96000),
                           (3, "enc_hilbert_6_4_7.bin", 48000), (4, "enc_hilbert_6_4_7.bin", 44100),
                           (5, "enc_hilbert_6_0_13.bin", 192000), (6, "enc_hilbert_6_0_13.bin", 8000), (6, "enc_hilbert_6_4_7.bin", 96000)):
        run_case(f"hilbert_f{fmt}_fs{fs}", fmt, dict(kind="file", name=fname), lcg(600, 2, seed=31 + fmt), 22, 24, 0, fs=fs,
                 random=0, dither=31, block=64, scratch=32, full=True, manifest=manifest)
    with open(os.path.join(OUT, "hilbert_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print(f"{len(manifest)} runtime cases, {len(VARIANTS)} encoder fixtures")


if __name__ == "__main__":
    main()
