#!/usr/bin/env python3
"""One core with N channel strands (gain, 2-section cascade, delay line, dithered store each: not a chain core):
dspRuntimeBlockAll with and without cutting the core into strand groups.  Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from avdsp_amd import encoder as enc, progbuilder as pb, runtime as rt
from tests.fuzz_programs import _prototypes

FPEAK, F48000 = 74, 5


def program(nch, fmt):
    def build(L):
        L.dsp_PARAM()
        banks = []
        for c in range(nch):
            b = L.dspBiquad_Sections(2)
            for k in range(2):
                L.dsp_Filter2ndOrder(FPEAK, 150.0 * (k + 1) + 7 * c, 1.0, 0.95)
            banks.append(b)
        L.dsp_CORE()
        L.dsp_TPDF_CALC(0)
        for c in range(nch):
            L.dsp_LOAD_GAIN_Fixed(128 + c, 0.5); L.dsp_GAIN_Fixed(0.9); L.dsp_BIQUADS(banks[c])
            L.dsp_DELAY_FixedMicroSec(100 + 10 * c); L.dsp_SAT0DB_TPDF(); L.dsp_STORE(c)
    L = enc.lib(); _prototypes(L)
    return enc.encode(build, 2 if fmt == 2 else 6, F48000, F48000, max_io=256, capacity=1 << 16)


for fmt in (2, 6):
    for nch in (16, 100):
        prog = program(nch, fmt)
        frames = 4096
        x = torch.from_numpy(pb.lcg_input(frames, nch, fmt == 6, seed=1)).cuda()
        y = torch.zeros((frames, nch), dtype=x.dtype, device="cuda")
        res = {}
        for split in (1, 0):
            r = rt.Runtime(fmt, prog, fs=48000, random=1, dither=24)
            r.set_option("strand_split", split)
            st = torch.cuda.current_stream().cuda_stream
            call = lambda: r._check(r.L.dspRuntimeBlockAllDevice(fmt, r.rundata, x.data_ptr(), nch, 128, y.data_ptr(), nch, 0, frames, st))
            for _ in range(2): call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): call()
            e1.record(); torch.cuda.synchronize()
            res[split] = (e0.elapsed_time(e1) * 1e3 / 5 / frames, r.get_option("pieces"), r.get_option("levels"))
            r.set_option("strand_split", 1); r.L.dspRuntimeRelease()
        print(f"fmt {fmt} {nch:4d} strands in one core: whole {res[0][0]:7.3f} us/frame; cut into {res[1][1]} pieces / {res[1][2]} levels "
              f"{res[1][0]:7.3f} us/frame ({res[0][0] / res[1][0]:.1f}x)", flush=True)
