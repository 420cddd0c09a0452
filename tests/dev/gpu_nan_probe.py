#!/usr/bin/env python3
"""Development probe (GPU box): where do NaN bit patterns differ between the device interpreter and the
oracle?  Prints the first frames of a small float-format program built around DSP_DITHER."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avdsp_amd import encoder as enc, progbuilder as pb, runtime as rt
from tests.fuzz_programs import _prototypes
from oracle import pyoracle as po


def build(L):
    _prototypes(L)
    L.dsp_CORE()
    L.dsp_TPDF_CALC(24)
    L.dsp_LOAD_GAIN_Fixed(32, 0.5)
    L.dsp_LOAD_GAIN_Fixed(33, 0.5)
    L.dsp_DITHER()
    L.dsp_STORE(0)
    L.dsp_SUBYX()
    L.dsp_SWAPXY()
    L.dsp_STORE(1)
    L.dsp_GAIN_Fixed(0.5)
    L.dsp_STORE(2)
    L.dsp_SAT0DB_TPDF_GAIN_Fixed(0.4)
    L.dsp_STORE(3)


for fmt in (5, 6, 3):
    prog = enc.encode(build, 6, 5, 5, max_io=48)
    x = pb.lcg_input(8, 2, fmt in (5, 6), seed=3)
    o = po.OracleProgram(fmt, prog, fs=48000, random=5, dither=24)
    r = rt.Runtime(fmt, prog, fs=48000, random=5, dither=24)
    want = o.run_block(x, 4, 32, 0, scratch_len=48, block=1)
    got = r.run_block(x, 4, 32, 0, block=1)
    print("fmt", fmt)
    for n in range(8):
        print("  frame", n, "gpu", [hex(v) for v in got[n].view(np.uint32)], "oracle", [hex(v) for v in want[n].view(np.uint32)])
    st = r.sync_state()
    print("  state gpu   ", [hex(v) for v in st[:12]])
    print("  state oracle", [hex(v) for v in o.state[:12]])
    r.release()
