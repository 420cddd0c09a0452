#!/usr/bin/env python3
"""Kernel time per frame of the general interpreter against the block length, both kernels.
Run on the GPU box:  python tools/interp_blocksize.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from avdsp_amd import progbuilder as pb, runtime as rt      # noqa: E402

prog = np.fromfile(os.path.join(ROOT, "tests", "golden", "crossoverLV6.bin"), dtype=np.uint32)
print("crossoverLV6.bin, DSP_FORMAT 2, 2 cores; kernel microseconds per frame (both cores)")
print(f"{'block':>6s} {'frame-parallel':>15s} {'frame by frame':>15s}")
for block in (1, 2, 4, 8, 16, 32, 64, 128, 256, 1024, 4096):
    frames = max(block * 4, 256)
    x = pb.lcg_input(frames, 16, False, seed=5)
    res = {}
    for impl in (1, 0):
        r = rt.Runtime(2, prog, fs=48000, random=1, dither=24)
        r.set_option("interp_impl", impl)
        r.run_block(x[:block], 32, 8)
        r.set_option("profile", 1)
        r.kernel_time(3); r.kernel_time(5)
        r.run_block(x, 32, 8, block=block)
        ms = r.kernel_time(3)[0] + r.kernel_time(5)[0]
        res[impl] = ms * 1e3 / frames
        r.set_option("profile", 0); r.set_option("interp_impl", 1)
        r.L.dspRuntimeRelease()
    print(f"{block:6d} {res[1]:15.2f} {res[0]:15.2f}", flush=True)
