"""Random valid DSP programs for differential testing (test infrastructure, not a test module).

`random_program(seed, fmt)` drives this repository's encoder library (byte-identical to the reference
encoder, tests/test_encoder.py) through a random but well-formed sequence of dsp_XXX() calls: several
cores, each a few "strands" that load, run a random selection of X/Y arithmetic, filters, delay lines,
meters, dither ... and store.  The same (seed, fmt) always gives the same words.

Used three ways:
  * tests/golden/make_goldens.py runs a set of seeds through the compiled reference and commits the
    outputs (tests/golden/fuzz_*.npz);
  * tests/test_oracle_golden.py holds the oracle to them (CPU);
  * tests/test_gpu_parity.py holds the general device interpreter to them (GPU).

Avoided on purpose, because the reference's result is not defined there: integer division by a runtime
value (traps on zero), DSP_FIR taps and the generators in the int64 model, square roots of negative
numbers and anything that can overflow a float accumulator into Inf/NaN (its -Ofast build assumes finite
math)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from avdsp_amd import encoder as enc

import os

NAN_HEAVY = bool(int(os.environ.get("AVDSP_FUZZ_NAN_HEAVY", "0")))
WIDE = bool(int(os.environ.get("AVDSP_FUZZ_WIDE", "0")))            # development switch: many strands per core (strand groups)
F44100, F48000, F96000, F192000 = 4, 5, 7, 9
N_IN, IN_BASE, N_OUT = 8, 32, 24           # inputs IO 32..39, outputs IO 0..23


class _Builder:
    def __init__(self, L, rng, fmt, fmin, fmax):
        self.L, self.r, self.fmt = L, rng, fmt
        self.int_mode = fmt == 2
        self.nf = fmax - fmin + 1
        self.fmin, self.fmax = fmin, fmax
        self.next_out = 0
        self.ns2_ok = fmin >= F44100 and fmax <= F192000

    # ---- parameters (one PARAM region in front of the cores) ----
    def params(self):
        L, r = self.L, self.r
        L.dsp_PARAM()
        self.banks = []
        for _ in range(3):
            n = int(r.integers(1, 4))
            b = L.dspBiquad_Sections(n)
            for _ in range(n):
                kind = int(r.choice([65, 67, 69, 71, 73, 74]))       # FLP2 FHP2 FLS2 FHS2 FAP2 FPEAK (enum filterTypes)
                L.dsp_Filter2ndOrder(kind, float(r.uniform(40, 8000)), float(r.uniform(0.4, 3.0)), float(r.uniform(0.5, 1.5)))
            self.banks.append(b)
        self.mux = L.dspLoadMux_Inputs(3)
        for _ in range(3):
            L.dspLoadMux_Data(IN_BASE + int(r.integers(0, N_IN)), float(r.uniform(-0.4, 0.4)))
        tab = (r.uniform(-0.9, 0.9, 16)).astype(np.float32)
        self.table = L.dspDataTableFloat(tab.ctypes.data_as(C.POINTER(C.c_float)), 16) if not self.int_mode else L.dspGenerator_Sine(16)
        if self.ns2_ok:
            c = (r.uniform(-0.6, 0.6, 3 * self.nf)).astype(np.float32)
            self.ns2 = L.dspDataTableFloat(c.ctypes.data_as(C.POINTER(C.c_float)), 3 * self.nf)
        self.mem = L.dspMem_LocationMultiple(2)
        self.delay = L.dspDelay_MicroSec_Max_Default(400, int(r.integers(20, 400)))
        # a plain default LAST: a float table at the end of the region would leave the reference encoder's
        # listing cursor on a data word, whose upper half then ends up in the header as "maxOpcode"
        # (dsp_encoder.c:296-297) and its own runtime refuses the program with -5
        self.gain = L.dspGain_Default(float(r.uniform(0.2, 0.9)))
        self.value = L.dspValue_Default(float(r.uniform(-0.5, 0.5)))

    # ---- one strand: load, a few operations, saturate, store ----
    def strand(self):
        L, r = self.L, self.r
        src = int(r.integers(0, 4))
        if src == 0:
            L.dsp_LOAD_GAIN_Fixed(IN_BASE + int(r.integers(0, N_IN)), float(r.uniform(0.1, 0.9)))
        elif src == 1:
            L.dsp_LOAD_GAIN(IN_BASE + int(r.integers(0, N_IN)), self.gain)
        elif src == 2:
            L.dsp_LOAD_MUX(self.mux)
        else:
            L.dsp_LOAD_GAIN_Fixed(IN_BASE + int(r.integers(0, N_IN)), 0.5)
            L.dsp_LOAD_GAIN_Fixed(IN_BASE + int(r.integers(0, N_IN)), 0.25)       # Y = first, X = second
        for _ in range(int(r.integers(1, 6))):
            self.operation()
        fin = int(r.integers(0, 5))
        if fin == 0: L.dsp_SAT0DB()
        elif fin == 1: L.dsp_SAT0DB_TPDF()
        elif fin == 2: L.dsp_SAT0DB_GAIN_Fixed(float(r.uniform(0.3, 1.2)))
        elif fin == 3: L.dsp_SAT0DB_TPDF_GAIN_Fixed(float(r.uniform(0.3, 1.2)))
        else: L.dsp_SAT0DB_TPDF_GAIN(self.gain)
        if r.random() < 0.3:
            L.dsp_DELAY(self.delay) if r.random() < 0.5 else L.dsp_DELAY_FixedMicroSec(int(r.integers(10, 300)))
        L.dsp_STORE(self.next_out % N_OUT)
        self.next_out += 1

    def operation(self):
        L, r = self.L, self.r
        ops = ["bq", "gain", "gainp", "copyxy", "swap", "addxy", "subxy", "subyx", "addyx", "avgxy", "avgyx", "negx", "negy",
               "copyyx", "delay1", "delaydp", "mulf", "divf", "muli_divi", "shift", "clip", "dcblock", "dither", "rms",
               "pwrxy", "mem", "value", "tpdf", "white_mix", "table"]
        if self.ns2_ok: ops.append("ns2")
        if not self.int_mode: ops += ["square_sqrt", "generator"]
        if NAN_HEAVY:                                   # development switch: DSP_DITHER manufactures NaN in the float models
            ops += ["dither"] * 8 + ["addxy", "subxy", "subyx", "addyx", "avgxy", "avgyx", "dcblock", "rms", "bq"] * 2
        op = str(r.choice(ops))
        if op == "bq": L.dsp_BIQUADS(self.banks[int(r.integers(0, len(self.banks)))])
        elif op == "gain": L.dsp_GAIN_Fixed(float(r.uniform(0.2, 1.0)))
        elif op == "gainp": L.dsp_GAIN(self.gain)
        elif op == "copyxy": L.dsp_COPYXY()
        elif op == "swap": L.dsp_SWAPXY()
        elif op == "addxy": L.dsp_AVGXY() if r.random() < 0.3 else (L.dsp_ADDXY(), L.dsp_GAIN_Fixed(0.5))
        elif op == "subxy": L.dsp_SUBXY(); L.dsp_GAIN_Fixed(0.5)
        elif op == "subyx": L.dsp_SUBYX(); L.dsp_SWAPXY(); L.dsp_GAIN_Fixed(0.5)
        elif op == "addyx": L.dsp_ADDYX(); L.dsp_SWAPXY(); L.dsp_GAIN_Fixed(0.5)
        elif op == "avgxy": L.dsp_AVGXY()
        elif op == "avgyx": L.dsp_AVGYX(); L.dsp_COPYYX()
        elif op == "negx": L.dsp_NEGX()
        elif op == "negy": L.dsp_NEGY()
        elif op == "copyyx": L.dsp_COPYYX()
        elif op == "delay1": L.dsp_DELAY_1()
        elif op == "delaydp": L.dsp_DELAY_DP_FixedMicroSec(int(r.integers(10, 200)))
        elif op == "mulf": L.dsp_MUL_Fixed(float(r.uniform(-0.9, 0.9))) if not self.int_mode else L.dsp_GAIN_Fixed(float(r.uniform(-0.9, 0.9)))
        elif op == "divf":
            if not self.int_mode: L.dsp_DIV_Fixed(float(r.choice([-1, 1]) * r.uniform(1.0, 3.0)))
        elif op == "muli_divi": L.dsp_MUL_FixedInt(int(r.integers(2, 9))); L.dsp_DIV_FixedInt(int(r.integers(9, 30)) * int(r.choice([-1, 1])))
        elif op == "shift":
            k = int(r.integers(1, 4)); L.dsp_SHIFT(-k)
            if r.random() < 0.5: L.dsp_SHIFT(k)
        elif op == "clip": L.dsp_CLIP_Fixed(float(r.uniform(0.05, 0.9)))
        elif op == "dcblock": L.dsp_DCBLOCK(int(r.integers(1, 60)))
        elif op == "dither": L.dsp_DITHER()
        elif op == "ns2": L.dsp_DITHER_NS2(self.ns2)
        elif op == "rms":
            L.dsp_SAT0DB(); L.dsp_RMS(int(r.choice([10, 20])), int(r.integers(0, 4)))
            if not self.int_mode: L.dsp_GAIN_Fixed(0.03125)         # float meters read sqrt(sum of squares), up to ~30
        elif op == "pwrxy":
            L.dsp_SAT0DB(); L.dsp_COPYXY(); L.dsp_PWRXY(10, int(r.integers(0, 3)))
            if not self.int_mode: L.dsp_GAIN_Fixed(0.03125)
        elif op == "mem":
            k = int(r.integers(0, 2)); L.dsp_STORE_MEM_Index(self.mem, k); L.dsp_LOAD_MEM_Index(self.mem, int(r.integers(0, 2)))
        elif op == "value": L.dsp_VALUE(self.value) if r.random() < 0.5 else L.dsp_VALUE_Fixed(float(r.uniform(-0.5, 0.5))); L.dsp_AVGXY()
        elif op == "tpdf": L.dsp_COPYXY(); L.dsp_TPDF(int(r.integers(8, 25))); L.dsp_SWAPXY()
        elif op == "white_mix":
            L.dsp_COPYXY(); L.dsp_WHITE()
            if self.int_mode: L.dsp_SHIFT(20)                      # s.31 noise up to the 5.59 scale, 8 bits down
            else: L.dsp_SHIFT(-8)
            L.dsp_AVGXY()
        elif op == "table":
            L.dsp_COPYXY(); L.dsp_DATA_TABLE(self.table, 0.5, int(r.integers(1, 4)), 16)
            if self.int_mode: pass
            L.dsp_AVGXY()
        elif op == "square_sqrt": L.dsp_COPYXY(); L.dsp_MULXY(); L.dsp_SQRTX()
        elif op == "generator":
            L.dsp_COPYXY()
            g = int(r.integers(0, 3))
            if g == 0: L.dsp_DIRAC_Fixed(int(r.integers(100, 2000)), 0.5)
            elif g == 1: L.dsp_SQUAREWAVE_Fixed(int(r.integers(100, 2000)), 0.5)
            else: L.dsp_SINE_Fixed(int(r.integers(100, 2000)), 0.5)
            L.dsp_AVGXY()

    def build(self):
        L, r = self.L, self.r
        self.params()
        for core in range(int(r.integers(1, 4))):
            L.dsp_CORE()
            if core == 0:
                L.dsp_TPDF_CALC(int(r.choice([0, 16, 24])))
            if r.random() < 0.3:
                L.dsp_LOAD_STORE()
                L.dspLoadStore_Data(IN_BASE + int(r.integers(0, N_IN)), 20 + int(r.integers(0, 4)))
            for _ in range(int(r.integers(6, 20)) if WIDE else int(r.integers(1, 5))):
                self.strand()
            if r.random() < 0.3:
                L.dsp_LOAD(IN_BASE + int(r.integers(0, N_IN)))
                L.dsp_DISTRIB(23, int(r.choice([8, 16, 32])))


def _prototypes(L):
    i32, f32, f64 = C.c_int, C.c_float, C.c_double
    for name, args in {
        "dsp_LOAD_GAIN": [i32, i32], "dsp_LOAD_MUX": [i32], "dspLoadMux_Inputs": [i32], "dspLoadMux_Data": [i32, f32],
        "dspDataTableFloat": [C.POINTER(f32), i32], "dspGenerator_Sine": [i32], "dspMem_LocationMultiple": [i32],
        "dspGain_Default": [f32], "dspValue_Default": [f32], "dspDelay_MicroSec_Max_Default": [i32, i32],
        "dsp_SAT0DB_TPDF_GAIN_Fixed": [f32], "dsp_SAT0DB_TPDF_GAIN": [i32], "dsp_DELAY": [i32], "dsp_GAIN": [i32],
        "dsp_DELAY_DP_FixedMicroSec": [i32], "dsp_MUL_Fixed": [f32], "dsp_DIV_Fixed": [f32], "dsp_MUL_FixedInt": [i32],
        "dsp_DIV_FixedInt": [i32], "dsp_SHIFT": [i32], "dsp_CLIP_Fixed": [f32], "dsp_DCBLOCK": [i32], "dsp_DITHER_NS2": [i32],
        "dsp_RMS": [i32, i32], "dsp_PWRXY": [i32, i32], "dsp_STORE_MEM_Index": [i32, i32], "dsp_LOAD_MEM_Index": [i32, i32],
        "dsp_VALUE": [i32], "dsp_VALUE_Fixed": [f32], "dsp_TPDF": [i32], "dsp_DATA_TABLE": [i32, f32, i32, i32],
        "dsp_DIRAC_Fixed": [i32, f32], "dsp_SQUAREWAVE_Fixed": [i32, f32], "dsp_SINE_Fixed": [i32, f32],
        "dspLoadStore_Data": [i32, i32], "dsp_DISTRIB": [i32, i32],
    }.items():
        getattr(L, name).argtypes = args


def random_program(seed: int, fmt: int, encoder_path: str | None = None) -> np.ndarray:
    """Program words for (seed, fmt).  fmt 2 -> Q28 encoding, 3..6 -> float encoding (one program per
    encoding: the float-encoded program of a seed is the same for formats 3, 4, 5 and 6).
    `encoder_path` swaps in another encoder library with the same API (the reference's, for comparison)."""
    L = enc.lib(encoder_path)
    _prototypes(L)
    rng = np.random.default_rng(1000 * seed + (2 if fmt == 2 else 6))
    fmin, fmax = [(F44100, F96000), (F48000, F48000), (F44100, F192000)][seed % 3]

    def build(lib):
        _Builder(lib, rng, 2 if fmt == 2 else 6, fmin, fmax).build()

    return enc.encode(build, 2 if fmt == 2 else 6, fmin, fmax, max_io=48, capacity=1 << 15, path=encoder_path)


def stress_input(rng, n: int, ch: int, float_samples: bool) -> np.ndarray:
    """Samples that visit the corners: exact and negative zeros, full scale, beyond full scale (float),
    barely-normal and subnormal magnitudes, small integers."""
    kind = rng.integers(0, 8, (n, ch))
    if float_samples:
        base = rng.uniform(-1, 1, (n, ch)).astype(np.float32)
        x = np.where(kind == 0, 0.0, base)
        x = np.where(kind == 1, base * 4.0, x)
        x = np.where(kind == 2, base * 1e-38, x)
        x = np.where(kind == 3, base * 1e-30, x)
        x = np.where(kind == 4, np.sign(base) * 1.0, x)
        x = np.where(kind == 5, -0.0, x)
        return x.astype(np.float32)
    base = rng.integers(-2**31, 2**31, (n, ch), dtype=np.int64)
    x = np.where(kind == 0, 0, base)
    x = np.where(kind == 1, base >> 20, x)
    x = np.where(kind == 2, np.where(base > 0, 2**31 - 1, -2**31), x)
    x = np.where(kind == 3, base >> 30, x)
    return x.astype(np.int32)


def random_chain_case(seed: int):
    """(channels, sections, taps, fmin, fmax, gain, fs, frames, dither, block) of a random chain program."""
    rng = np.random.default_rng(seed)
    C, S, T = int(rng.integers(1, 5)), int(rng.integers(0, 6)), int(rng.choice([0, 0, 1, 2, 7, 33, 100]))
    fmin = int(rng.integers(4, 8)); fmax = int(rng.integers(fmin, min(fmin + 3, 10)))
    gain = float(rng.choice([0.1, 0.5, 1.0, 2.0, 7.9]))
    fs = [8000, 16000, 24000, 32000, 44100, 48000, 88200, 96000, 176400, 192000][int(rng.integers(fmin, fmax + 1))]
    n = int(rng.integers(50, 400)); dither = int(rng.choice([16, 24, 31]))
    return rng, C, S, T, fmin, fmax, gain, fs, n, dither
