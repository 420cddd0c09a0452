#!/usr/bin/env python3
"""One core with N channel strands (gain, 2-section cascade, delay line, dithered store each: not a chain core):
dspRuntimeBlockAll with and without cutting the core into strand groups.  Run on the GPU box."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from avdsp_amd import encoder as enc, progbuilder as pb, runtime as rt
from avdsp_amd import devmem as dm
from tests.fuzz_programs import _prototypes

FPEAK, F48000 = 74, 5


def program(nch, fmt, in_base=128):
    def build(L):
        banks = []
        for c in range(nch):
            if c % 64 == 0:
                L.dsp_PARAM()                                   # (a PARAM section's length is a 16-bit word)
            b = L.dspBiquad_Sections(2)
            for k in range(2):
                L.dsp_Filter2ndOrder(FPEAK, 150.0 * (k + 1) + 7 * c, 1.0, 0.95)
            banks.append(b)
        L.dsp_CORE()
        L.dsp_TPDF_CALC(0)
        for c in range(nch):
            L.dsp_LOAD_GAIN_Fixed(in_base + c, 0.5); L.dsp_GAIN_Fixed(0.9); L.dsp_BIQUADS(banks[c])
            L.dsp_DELAY_FixedMicroSec(100 + 10 * c); L.dsp_SAT0DB_TPDF(); L.dsp_STORE(c)
    L = enc.lib(); _prototypes(L)
    return enc.encode(build, 2 if fmt == 2 else 6, F48000, F48000, max_io=max(256, in_base + nch), capacity=max(1 << 16, 64 * nch))


import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--strands", type=int, nargs="*", default=[16, 100, 4096])
ap.add_argument("--fmt", type=int, nargs="*", default=[2, 6])
ap.add_argument("--lanes-only", action="store_true", help="only the default path (the run of strands on lanes): for a profiler")
args = ap.parse_args()

for fmt in args.fmt:
    for nch in args.strands:
        in_base = max(128, nch)
        prog = program(nch, fmt, in_base)
        frames = 4096 if nch <= 100 else 1024
        x = dm.to_device(pb.lcg_input(frames, nch, fmt == 6, seed=1))
        y = torch.zeros((frames, nch), dtype=x.dtype, device="cuda")
        res = {}
        for split in (2, 1, 0):                                   # 2: strand runs on lanes (the default); 1: strand groups through the interpreter; 0: the core whole
            if (nch > 100 and split == 0) or (args.lanes_only and split != 2):
                res[split] = (float("nan"), 0, 0, 0); continue
            r = rt.Runtime(fmt, prog, fs=48000, random=1, dither=24)
            r.set_option("strand_split", 1 if split else 0)
            r.set_option("strand_lanes", 2 if split == 2 else 0)           # 2: lower the run whatever its length (the default only beyond 64 strands)
            st = torch.cuda.current_stream().cuda_stream
            call = lambda: r._check(r.L.dspRuntimeBlockAllDevice(fmt, r.rundata, x.data_ptr(), nch, in_base, y.data_ptr(), nch, 0, frames, st))
            for _ in range(2): call()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): call()
            e1.record(); torch.cuda.synchronize()
            r.set_option("profile", 1)
            for k in (3, 5, 6): r.kernel_time(k)
            call(); torch.cuda.synchronize()
            kt = {k: r.kernel_time(k)[0] * 1e3 / frames for k in (3, 5, 6)}
            r.set_option("profile", 0)
            if split == 2:
                print(f"    kernels per frame: interpreter frame by frame {kt[3]:.3f} us, frame-parallel {kt[5]:.3f} us, strand_lanes {kt[6]:.3f} us")
            res[split] = (e0.elapsed_time(e1) * 1e3 / 5 / frames, r.get_option("pieces"), r.get_option("levels"), r.get_option("strands"))
            r.set_option("strand_split", 1); r.set_option("strand_lanes", 1); r.L.dspRuntimeRelease()
        print(f"fmt {fmt} {nch:4d} strands in one core: whole {res[0][0]:7.3f} us/frame; cut into {res[1][1]} pieces / {res[1][2]} levels "
              f"{res[1][0]:7.3f} us/frame; {res[2][3]} strands on lanes {res[2][0]:7.3f} us/frame = {nch / res[2][0] / 1e3:.2f} Gsamples/s", flush=True)
