#!/usr/bin/env python3
"""Per-opcode cost of the general interpreter, by differencing: a core of LOAD ... STORE with K copies of
one opcode in between, K = 0 and K = 32, both interpreter kernels (frame-parallel / frame by frame).
Run on the GPU box:  python tools/interp_opcost.py [fmt]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from avdsp_amd import encoder as enc, progbuilder as pb, runtime as rt      # noqa: E402
from tests.fuzz_programs import _prototypes                                   # noqa: E402

F48000, IN, FPEAK = 5, 32, 74
FRAMES = 64 * 256


def program(fmt, op, k):
    taps = np.linspace(-0.1, 0.1, 64).astype(np.float32)

    def build(L):
        L.dsp_PARAM()
        banks = {}
        for n in (1, 6, 16, 64):
            banks[n] = L.dspBiquad_Sections(n)
            for j in range(n):
                L.dsp_Filter2ndOrder(FPEAK, 200.0 + 90.0 * j, 0.9, 1.0)
        fir = 0
        if fmt != 2:
            fir = L.dspFir_Impulses()
            L.dspFir_ImpulseData(taps.ctypes.data_as(C.POINTER(C.c_float)), 64)
        mux = L.dspLoadMux_Inputs(4)
        for j in range(4):
            L.dspLoadMux_Data(IN + j, 0.2)
        L.dsp_CORE()
        L.dsp_TPDF_CALC(0)
        L.dsp_LOAD_GAIN_Fixed(IN, 0.5)
        L.dsp_DELAY_1()                                   # keeps the core out of the chain kernels
        for _ in range(k):
            if op == "gain": L.dsp_GAIN_Fixed(0.99)
            elif op == "swapxy": L.dsp_SWAPXY()
            elif op == "store": L.dsp_STORE(1)
            elif op == "load_gain": L.dsp_LOAD_GAIN_Fixed(IN + 1, 0.5)
            elif op == "load_mux4": L.dsp_LOAD_MUX(mux)
            elif op == "sat_tpdf_gain": L.dsp_SAT0DB_TPDF_GAIN_Fixed(0.9)
            elif op == "delay_1": L.dsp_DELAY_1()
            elif op == "delay_100": L.dsp_DELAY_FixedMicroSec(2084)
            elif op == "delay_dp_100": L.dsp_DELAY_DP_FixedMicroSec(2084)
            elif op == "dcblock": L.dsp_DCBLOCK(10)
            elif op == "rms": L.dsp_RMS(10, 2)
            elif op == "dither": L.dsp_DITHER()
            elif op == "fir64": L.dsp_FIR(fir)
            elif op.startswith("biquads"): L.dsp_BIQUADS(banks[int(op[7:])])
        L.dsp_SAT0DB()
        L.dsp_STORE(0)

    L = enc.lib()
    _prototypes(L)
    return enc.encode(build, 2 if fmt == 2 else 6, F48000, F48000, max_io=48, capacity=1 << 16)


def time_us_per_frame(fmt, prog, impl, x):
    r = rt.Runtime(fmt, prog, fs=48000, random=1, dither=24)
    r.set_option("interp_impl", impl)
    r.run_block(x[:64], 8, IN)
    r.set_option("profile", 1)
    r.kernel_time(3); r.kernel_time(5)
    r.run_block(x, 8, IN)
    ms = r.kernel_time(3)[0] + r.kernel_time(5)[0]
    r.set_option("profile", 0); r.set_option("interp_impl", 1)
    r.L.dspRuntimeRelease()
    return ms * 1e3 / len(x)


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    fmt = int(argv[0]) if argv else 6
    only = [a[7:].split(",") for a in sys.argv[1:] if a.startswith("--only=")]       # --only=biquads1,biquads6
    wave_only = "--wave-only" in sys.argv                                             # skip the (slow) frame-by-frame kernel
    ops = ["gain", "swapxy", "store", "load_gain", "load_mux4", "sat_tpdf_gain", "delay_1", "delay_100", "delay_dp_100",
           "dcblock", "rms", "dither", "biquads1", "biquads6", "biquads16", "biquads64"] + (["fir64"] if fmt != 2 else [])
    if only: ops = [o for o in ops if o in only[0]]
    impls = (1,) if wave_only else (1, 0)
    x = pb.lcg_input(FRAMES, 8, fmt in (5, 6), seed=5)
    K = 32
    print(f"DSP_FORMAT {fmt}, {FRAMES} frames per block; ns per opcode and frame")
    print(f"{'opcode':16s} {'frame-parallel':>15s} {'frame by frame':>15s} {'ratio':>7s}")
    base = {impl: time_us_per_frame(fmt, program(fmt, "none", 0), impl, x if impl else x[:4096]) for impl in impls}
    base.setdefault(0, float("nan"))
    print(f"{'(empty core)':16s} {base[1] * 1e3:15.1f} {base[0] * 1e3:15.1f}")
    for op in ops:
        prog = program(fmt, op, K)
        c = {impl: (time_us_per_frame(fmt, prog, impl, x if impl else x[:4096]) - base[impl]) / K * 1e3 for impl in impls}
        c.setdefault(0, float("nan"))
        print(f"{op:16s} {c[1]:15.1f} {c[0]:15.1f} {c[0] / max(c[1], 1e-9):7.1f}", flush=True)


if __name__ == "__main__":
    main()
