"""Strand plans (include/avdsp_hip.h): a stretch of an interpreted core that is N repetitions of one opcode sequence runs with
lane = strand (strand_lanes) instead of one wave per strand group.  Every case is held to the oracle bit for bit -- outputs and
the final state area -- over several blocks of awkward sizes, with the lowering on and off, and the programs the reference ships
(whose runs the host finds by itself) to the reference's own goldens through the existing golden tests."""
import os

import numpy as np
import pytest

from avdsp_amd import encoder as enc
from avdsp_amd import progbuilder as pb
from avdsp_amd import runtime as rt
from oracle import pyoracle as po
from tests.fuzz_programs import _prototypes

pytestmark = pytest.mark.gpu
FPEAK, FLP2, F48000 = 74, 65, 5


@pytest.fixture(autouse=True)
def _release():
    yield
    rt.lib().dspRuntimeRelease()


def words(a):
    return np.ascontiguousarray(a).view(np.uint32)


def crossover_program(nch, fmt, shape):
    """one core: [TPDF_CALC] + nch strands of `shape`; inputs at IO 128.., outputs from IO 0"""
    def build(L):
        L.dsp_PARAM()
        banks, banks2 = [], []
        for c in range(nch):
            b = L.dspBiquad_Sections(2)
            for k in range(2):
                L.dsp_Filter2ndOrder(FPEAK, 150.0 * (k + 1) + 7 * c, 1.0, 0.95)
            banks.append(b)
            b = L.dspBiquad_Sections(3)
            for k in range(3):
                L.dsp_Filter2ndOrder(FLP2, 900.0 + 31 * c + 100 * k, 0.7, 1.0)
            banks2.append(b)
        L.dsp_CORE()
        if shape != "plain":
            L.dsp_TPDF_CALC(0)
        for c in range(nch):
            if shape == "plain":                              # gain, cascade, delay, store
                L.dsp_LOAD_GAIN_Fixed(128 + c, 0.5); L.dsp_GAIN_Fixed(0.9); L.dsp_BIQUADS(banks[c])
                L.dsp_DELAY_FixedMicroSec(30 + 17 * c); L.dsp_SAT0DB(); L.dsp_STORE(c)
            elif shape == "dither":                           # tools/wide_core_bench.py's strand
                L.dsp_LOAD_GAIN_Fixed(128 + c, 0.5); L.dsp_GAIN_Fixed(0.9); L.dsp_BIQUADS(banks[c])
                L.dsp_DELAY_FixedMicroSec(100 + 10 * c); L.dsp_SAT0DB_TPDF(); L.dsp_STORE(c)
            elif shape == "subtractive":                      # crossoverLV6's core 1: delayed minus filtered, two outputs
                L.dsp_LOAD(128 + c); L.dsp_COPYXY(); L.dsp_DELAY_FixedMicroSec(200 + 3 * c); L.dsp_GAIN_Fixed(1.0); L.dsp_SWAPXY()
                L.dsp_GAIN_Fixed(0.8); L.dsp_BIQUADS(banks2[c]); L.dsp_SUBYX(); L.dsp_SAT0DB_TPDF(); L.dsp_STORE(2 * c)
                L.dsp_SWAPXY(); L.dsp_SAT0DB_TPDF(); L.dsp_STORE(2 * c + 1)
            elif shape == "two_way":                          # crossoverLV6's core 2: one input, two banks, two outputs
                L.dsp_LOAD_GAIN_Fixed(128 + c, 0.7); L.dsp_COPYXY(); L.dsp_BIQUADS(banks[c]); L.dsp_SAT0DB_TPDF_GAIN_Fixed(0.9); L.dsp_STORE(2 * c)
                L.dsp_SWAPXY(); L.dsp_BIQUADS(banks2[c]); L.dsp_DELAY_DP_FixedMicroSec(40 + 11 * c); L.dsp_SHIFT(-1)
                L.dsp_SAT0DB_GAIN_Fixed(1.1); L.dsp_STORE(2 * c + 1)
    L = enc.lib(); _prototypes(L)
    L.dsp_SAT0DB_TPDF_GAIN_Fixed.argtypes = [enc.C.c_float]
    return enc.encode(build, 2 if fmt == 2 else 6, F48000, F48000, max_io=256, capacity=1 << 17)


@pytest.mark.parametrize("fmt", [2, 3, 4, 5, 6])
@pytest.mark.parametrize("shape,nch", [("plain", 5), ("dither", 100), ("subtractive", 37), ("two_way", 64)])
def test_strand_run_matches_the_oracle(fmt, shape, nch):
    prog = crossover_program(nch, fmt, shape)
    nout = nch if shape in ("plain", "dither") else 2 * nch
    blocks = [1, 7, 8, 64, 100, 333]
    x = pb.lcg_input(sum(blocks), nch, fmt in (5, 6), seed=3 + nch)
    o = po.OracleProgram(fmt, prog, fs=48000, random=1, dither=24)
    want = np.concatenate([o.run_block(x[a:a + n], nout, 128) for a, n in zip(np.cumsum([0] + blocks[:-1]), blocks)])
    for lanes in (2, 0):                             # 2: every run on lanes (the default keeps runs of up to 64 strands with the interpreter)
        r = rt.Runtime(fmt, prog, fs=48000, random=1, dither=24)
        r.set_option("strand_lanes", lanes)
        got = np.concatenate([r.run_block_all(x[a:a + n], nout, 128) for a, n in zip(np.cumsum([0] + blocks[:-1]), blocks)])
        assert r.get_option("strands") == (nch if lanes else 0)
        bad = np.nonzero((words(got) != words(want)).any(axis=0))[0]
        assert bad.size == 0, f"lanes={lanes}: outputs {bad[:8].tolist()} differ, first frame {np.nonzero(words(got)[:, bad[0]] != words(want)[:, bad[0]])[0][:3].tolist()}"
        assert (r.sync_state() == o.state).all(), f"lanes={lanes}: state"
        r.set_option("strand_lanes", 1)
        r.release()


def test_per_core_entry_point_and_window_fallback():
    """dspRuntimeBlock_N on the core takes the same arrangement; windows that share IO numbers go through the interpreter"""
    fmt, nch = 6, 12
    prog = crossover_program(nch, fmt, "dither")
    x = pb.lcg_input(200, nch, True, seed=9)
    o = po.OracleProgram(fmt, prog, fs=48000, random=1, dither=24)
    want = o.run_block(x, nch, 128)
    r = rt.Runtime(fmt, prog, fs=48000, random=1, dither=24)
    r.set_option("strand_lanes", 2)
    got = r.run_block(x, nch, 128)
    assert r.get_option("strands") == nch
    assert (words(got) == words(want)).all()
    # one window for both: inputs at IO 128.., outputs at IO 0.. inside a 140-wide row
    o2 = po.OracleProgram(fmt, prog, fs=48000, random=1, dither=24)
    r2 = rt.Runtime(fmt, prog, fs=48000, random=1, dither=24)
    row = np.zeros((200, 140), dtype=np.float32); row[:, 128:] = x
    want2 = o2.run_block(row, 140, 0, 0, scratch_len=141)
    got2 = r2.run_block(row, 140, 0, 0)
    assert (words(got2) == words(want2)).all()
    assert (r2.sync_state() == o2.state).all()
    r2.set_option("strand_lanes", 1)


def test_reference_programs_find_their_runs():
    """dacdiy1.bin: every one of its four cores is [prefix +] two strands of one shape (the goldens hold the results: test_gpu_wave)"""
    prog = np.fromfile(os.path.join(os.path.dirname(__file__), "golden", "dacdiy1.bin"), dtype=np.uint32)
    x = pb.lcg_input(256, 16, False, seed=5)
    o = po.OracleProgram(2, prog, fs=48000, random=1, dither=24)
    want = o.run_block(x, 32, 8, 0)
    for lanes, strands in ((2, 6), (1, 0)):        # short runs stay with the interpreter unless asked for ("strand_lanes" 2)
        o = po.OracleProgram(2, prog, fs=48000, random=1, dither=24)
        want = o.run_block(x, 32, 8, 0)
        r = rt.Runtime(2, prog, fs=48000, random=1, dither=24)
        r.set_option("strand_lanes", lanes)
        try:
            got = r.run_block_all(x, 32, 8, 0)
            assert r.get_option("strands") == strands      # (the two strands of its last core store the same IOs: they meet, the interpreter keeps them)
            assert (words(got) == words(want)).all()
            assert (r.sync_state() == o.state).all()
        finally:
            r.set_option("strand_lanes", 1)
            r.release()
