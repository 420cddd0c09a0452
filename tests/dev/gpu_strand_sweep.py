#!/usr/bin/env python3
"""Development sweep (GPU box): cores that end in a run of N identical strands of a RANDOM shape -- loads, X/Y moves and sums, gains,
shifts, cascades, delay lines, saturation with and without dither, stores, memories -- through the strand plan (strand_lanes) and, for
comparison, the interpreter's strand groups, against the oracle: outputs and state, ragged blocks.
    python tests/dev/gpu_strand_sweep.py SEED0 SEED1"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from avdsp_amd import encoder as enc, progbuilder as pb, runtime as rt
from oracle import pyoracle as po
from tests.fuzz_programs import _prototypes

FPEAK, FLP2, F48000 = 74, 2, 5
IN0 = 512


def shape_of(rng):
    """one strand as a list of (name, argument kind); X/Y use is kept legal for a cut in front of every strand: the first opcode
    replaces X, and Y is only read after the strand itself has set it"""
    ops = [str(rng.choice(["load", "load_gain", "load_gain", "load_mem"]))]       # load_mem: a word a core in front of this one stores
    y_set = True                                            # LOAD / LOAD_GAIN leave the old X in Y -- but that is the strand's own only after a COPYXY
    y_own = False
    nst = 0
    for _ in range(int(rng.integers(2, 9))):
        pool = ["gain", "biquads", "delay", "sat", "sat_tpdf", "sat_gain", "sat_tpdf_gain", "shift", "negx", "copyxy", "store", "store_mem"]
        if y_own:
            pool += ["swapxy", "addxy", "subxy", "addyx", "subyx", "copyyx"]
        op = str(rng.choice(pool))
        if op == "copyxy":
            y_own = True
        if op == "store":
            nst += 1
            if nst > 3: continue
        ops.append(op)
    ops.append("sat_tpdf" if rng.random() < 0.5 else "sat")
    ops.append("store")
    return ops


def program(nch, fmt, ops, rng):
    nstore = sum(1 for o in ops if o == "store")
    shifts = [int(rng.integers(-3, 3)) for _ in ops]
    gains = [float(rng.uniform(0.3, 1.2)) for _ in ops]
    delays = [int(rng.integers(0, 700)) for _ in ops]
    nbq = [int(rng.integers(1, 4)) for _ in ops]
    dp = [bool(rng.random() < 0.3) for _ in ops]

    def build(L):
        banks = {}
        mem_in, mem_out = [], {}
        for c in range(nch):
            if c % 32 == 0:
                L.dsp_PARAM()
            mem_in.append(L.dspMem_Location())
            for i, o in enumerate(ops):
                if o == "store_mem": mem_out[(c, i)] = L.dspMem_Location()
            for i, o in enumerate(ops):
                if o == "biquads":
                    b = L.dspBiquad_Sections(nbq[i])
                    for k in range(nbq[i]):
                        L.dsp_Filter2ndOrder(FPEAK if k % 2 == 0 else FLP2, 120.0 * (k + 1) + 5 * c + 40 * i, 0.8, 0.9)
                    banks[(c, i)] = b
        if ops[0] == "load_mem":                              # a core that fills the memories the strands start from
            L.dsp_CORE()
            for c in range(nch):
                L.dsp_LOAD_GAIN_Fixed(IN0 + c, 0.7); L.dsp_STORE_MEM(mem_in[c])
        L.dsp_CORE()
        if any(o in ("sat_tpdf", "sat_tpdf_gain") for o in ops):
            L.dsp_TPDF_CALC(0)
        for c in range(nch):
            k = 0
            for i, o in enumerate(ops):
                if o == "load": L.dsp_LOAD(IN0 + c)
                elif o == "load_mem": L.dsp_LOAD_MEM(mem_in[c])
                elif o == "store_mem": L.dsp_STORE_MEM(mem_out[(c, i)])
                elif o == "load_gain": L.dsp_LOAD_GAIN_Fixed(IN0 + c, gains[i])
                elif o == "gain": L.dsp_GAIN_Fixed(gains[i])
                elif o == "biquads": L.dsp_BIQUADS(banks[(c, i)])
                elif o == "delay":
                    (L.dsp_DELAY_DP_FixedMicroSec if dp[i] else L.dsp_DELAY_FixedMicroSec)(delays[i] + 7 * c)
                elif o == "sat": L.dsp_SAT0DB()
                elif o == "sat_tpdf": L.dsp_SAT0DB_TPDF()
                elif o == "sat_gain": L.dsp_SAT0DB_GAIN_Fixed(gains[i])
                elif o == "sat_tpdf_gain": L.dsp_SAT0DB_TPDF_GAIN_Fixed(gains[i])
                elif o == "shift": L.dsp_SHIFT(shifts[i])
                elif o == "negx": L.dsp_NEGX()
                elif o == "copyxy": L.dsp_COPYXY()
                elif o == "copyyx": L.dsp_COPYYX()
                elif o == "swapxy": L.dsp_SWAPXY()
                elif o == "addxy": L.dsp_ADDXY()
                elif o == "addyx": L.dsp_ADDYX()
                elif o == "subxy": L.dsp_SUBXY()
                elif o == "subyx": L.dsp_SUBYX()
                elif o == "store":
                    L.dsp_STORE(nstore * c + k); k += 1
    L = enc.lib(); _prototypes(L)
    L.dsp_SAT0DB_TPDF_GAIN_Fixed.argtypes = [enc.C.c_float]
    return enc.encode(build, 2 if fmt == 2 else 6, F48000, F48000, max_io=IN0 + nch + 8, capacity=1 << 18), nstore


def run(lo, hi, formats=(2, 3, 4, 5, 6)):
    """seeds lo .. hi-1; returns (runs, list of mismatch descriptions, cases lowered to strand plans)"""
    bad, n, lowered = [], 0, 0
    for seed in range(lo, hi):
        rng = np.random.default_rng(seed)
        ops = shape_of(rng)
        nch = int(rng.choice([2, 3, 5, 16, 64, 65, 100]))
        for fmt in formats:
            rs = np.random.default_rng(seed * 7 + 1)
            try:
                prog, nstore = program(nch, fmt, ops, rs)
            except Exception as e:                              # (a shape the encoder refuses)
                print("seed", seed, "fmt", fmt, "encoder:", str(e)[:80]); break
            blocks = [int(b) for b in rs.choice([1, 7, 16, 33, 64, 100, 200], 3)]
            x = pb.lcg_input(sum(blocks), nch, fmt in (5, 6), seed=seed)
            nout = max(nstore * nch, 1)
            o = po.OracleProgram(fmt, prog, fs=48000, random=seed, dither=24)
            if o.rc < 0: break
            want = np.concatenate([o.run_block(x[a:a + b], nout, IN0) for a, b in zip(np.cumsum([0] + blocks[:-1]), blocks)])
            for lanes in (2, 0):
                r = rt.Runtime(fmt, prog, fs=48000, random=seed, dither=24)
                r.set_option("strand_lanes", lanes)
                try:
                    got = np.concatenate([r.run_block_all(x[a:a + b], nout, IN0) for a, b in zip(np.cumsum([0] + blocks[:-1]), blocks)])
                    ok = bool((got.view(np.uint32) == want.view(np.uint32)).all()) and bool((r.sync_state() == o.state).all())
                    if lanes == 2: lowered += r.get_option("strands") > 0
                except rt.AvdspError as e:
                    ok = False; print("   error:", e)
                n += 1
                if not ok:
                    bad.append(f"seed {seed} fmt {fmt} lanes {lanes} strands {nch} ops {ops}")
                r.set_option("strand_lanes", 1)
                r.release()
    return n, bad, lowered


if __name__ == "__main__":
    n, bad, lowered = run(int(sys.argv[1]), int(sys.argv[2]))
    for b in bad: print("MISMATCH", b)
    print("runs", n, "bad", len(bad), "cases lowered to strand plans", lowered)
