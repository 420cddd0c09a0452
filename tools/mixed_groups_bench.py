#!/usr/bin/env python3
"""A chain core whose chains have DIFFERENT section counts (a real crossover: 2 biquads on one way, 6 on another ...) is one cascade launch per
section count.  How long does a block take?  (round 5)   python tools/mixed_groups_bench.py"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from avdsp_amd import runtime as rt, encoder as enc, progbuilder as pb
from avdsp_amd import devmem as dm

def program(counts, per):
    """len(counts) * per chains: chain c has counts[c % len(counts)] peaking filters; inputs IO n .. 2n-1, outputs 0 .. n-1"""
    n = len(counts) * per
    def build(L):
        import ctypes as C
        banks = []
        for c in range(n):
            L.dsp_PARAM()
            banks.append(L.dspBiquad_Sections(counts[c % len(counts)]))
            for b in range(counts[c % len(counts)]):
                L.dsp_Filter2ndOrder(9, C.c_double(100.0 + 37 * b + 3 * (c % 97)), C.c_double(0.7 + 0.05 * (b % 5)), C.c_float(1.2 if b & 1 else 0.8))   # FPEAK
        L.dsp_CORE()
        for c in range(n):
            L.dsp_LOAD(n + c); L.dsp_BIQUADS(banks[c]); L.dsp_SAT0DB(); L.dsp_STORE(c)
    return enc.encode(build, 6, pb.F48000, pb.F48000, max_io=2 * n + 8, capacity=1 << 22), n

for counts, per in (([8], 1024), ([2, 4, 6, 8], 256), ([1, 2, 3, 4, 5, 6, 7, 8], 128), ([2, 4, 6, 8], 1024), ([16], 4096), ([4, 8, 12, 16], 1024)):
    prog, n = program(counts, per)
    r = rt.Runtime(6, prog)
    B = 1024
    x = dm.to_device(pb.lcg_input(B, n, True, seed=1)); y = torch.zeros((B, n), dtype=x.dtype, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(5): r.run_block_device(x.data_ptr(), n, n, y.data_ptr(), n, 0, B, st)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): r.run_block_device(x.data_ptr(), n, n, y.data_ptr(), n, 0, B, st)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print(f"{n:5d} chains, section counts {counts}: {dt * 1e6:8.1f} us per 1024-frame block ({len(counts)} section counts; round 4: as many cascade launches, one after the other)", flush=True)
    r.release()
