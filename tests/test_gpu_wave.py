"""The frame-parallel form of the general interpreter (avdsp_interp.inc, interp_wave: 64 frames of a block
side by side, one per lane) against the oracle: the forms that differ from the frame-by-frame code
(delay lines of every length around the batch size, DELAY_1, DELAY_DP, cascades longer than a wave,
DSP_FIR, per-lane memories, the TPDF sequence), block lengths around the batch size, and the host's
eligibility analysis (avdsp_host.c scan_generic): cores that hand a value from one frame to the next
must take the frame-by-frame kernel, and both must give the reference's result."""
import ctypes as C

import numpy as np
import pytest

from avdsp_amd import encoder as enc, progbuilder as pb, runtime as rt
from avdsp_amd import devmem as dm
from oracle import pyoracle as po
from tests.fuzz_programs import _prototypes

pytestmark = pytest.mark.gpu

F48000 = 5
IN = 32                     # input window: IO 32..39
DSP_LOAD_MEM_DATA = 60      # dsp_header.h:126 (no emitter in the reference's encoder API)
FPEAK = 74
KIND_SCALAR, KIND_WAVE = 3, 5


def encode(build, fmt, max_io=48):
    L = enc.lib()
    _prototypes(L)
    return enc.encode(lambda lib: build(lib), 2 if fmt == 2 else 6, F48000, F48000, max_io=max_io, capacity=1 << 15)


def run_both(fmt, prog, x, out_stride, in_base, out_base, block, expect_wave, span=48, impl=1):
    """device vs oracle, both with a persistent samples[] frame; returns (wave launches, scalar launches)"""
    o = po.OracleProgram(fmt, prog, fs=48000, random=11, dither=24)
    r = rt.Runtime(fmt, prog, fs=48000, random=11, dither=24)
    assert r.rc == o.rc and r.rc > 0
    r.set_option("interp_impl", impl)
    r.set_option("profile", 1)
    try:
        frame = np.zeros(max(span, 4096), dtype=np.uint32)
        want = o.run_block(x, out_stride, in_base, out_base, block=block, frame=frame)
        got = r.run_block(x, out_stride, in_base, out_base, block=block)
        wave, scalar = r.kernel_time(KIND_WAVE)[1], r.kernel_time(KIND_SCALAR)[1]
        bad = np.nonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=0))[0]
        first = np.nonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=1))[0]
        assert bad.size == 0, f"DSP_FORMAT {fmt} block {block}: output columns {list(bad)} differ from frame {first[0]}"
        r.sync_state()
        n = int(prog[1]) + int(prog[2])
        words = np.nonzero(r.buf[12:n] != o.buf[12:n])[0] + 12
        assert words.size == 0, f"DSP_FORMAT {fmt} block {block}: buffer words {list(words[:8])} differ after the run"
        if expect_wave is True:
            assert wave > 0 and scalar == 0, (wave, scalar)
        elif expect_wave is False:
            assert wave == 0 and scalar > 0, (wave, scalar)
        return wave, scalar
    finally:
        r.set_option("profile", 0)
        r.set_option("interp_impl", 1)
        r.release()


def forms_program(fmt):
    rng = np.random.default_rng(7)
    taps = {n: rng.uniform(-0.2, 0.2, n).astype(np.float32) for n in (1, 5, 64, 100, 300)}

    def build(L):
        L.dsp_PARAM()
        big = L.dspBiquad_Sections(70)                      # longer than a wave: two passes of the section pipeline
        for k in range(70):
            L.dsp_Filter2ndOrder(FPEAK, 100.0 + 97.0 * k, 0.7 + 0.01 * k, 1.0 + 0.002 * (k % 5 - 2))
        small = L.dspBiquad_Sections(3)
        for k in range(3):
            L.dsp_Filter2ndOrder(FPEAK, 300.0 * (k + 1), 1.2, 0.9)
        dly = L.dspDelay_MicroSec_Max_Default(5000, 4167)   # 200 samples
        firs = {}
        if fmt != 2:
            for n, t in taps.items():
                firs[n] = L.dspFir_Impulses()
                L.dspFir_ImpulseData(t.ctypes.data_as(C.POINTER(C.c_float)), n)
        fdel = L.dspFir_Impulses()
        L.dspFir_Delay(10)                                   # DSP_FIR used as a plain delay (:940-955)
        mux = L.dspLoadMux_Inputs(2)
        L.dspLoadMux_Data(IN + 0, 0.3)
        L.dspLoadMux_Data(IN + 7, -0.3)
        mem = L.dspMem_LocationMultiple(2)

        L.dsp_CORE()
        L.dsp_TPDF_CALC(0)
        for k, us in enumerate((21, 1312, 1334, 1355)):      # 1, 62, 64, 65 samples at 48 kHz
            L.dsp_LOAD_GAIN_Fixed(IN + k, 0.5); L.dsp_DELAY_FixedMicroSec(us); L.dsp_STORE(k)
        L.dsp_LOAD(IN + 4); L.dsp_DELAY(dly); L.dsp_SAT0DB_TPDF(); L.dsp_STORE(4)
        L.dsp_LOAD(IN + 5)
        for us in (21, 1334, 2100):
            L.dsp_DELAY_DP_FixedMicroSec(us)
        L.dsp_STORE(5)
        L.dsp_LOAD_GAIN_Fixed(IN + 6, 0.25); L.dsp_BIQUADS(big); L.dsp_SAT0DB(); L.dsp_STORE(6)
        L.dsp_LOAD(IN + 7); L.dsp_DELAY_1(); L.dsp_DELAY_1(); L.dsp_BIQUADS(small); L.dsp_SAT0DB(); L.dsp_STORE(7)
        mux_result = L.dsp_LOAD_MUX(mux)
        L.dsp_STORE_MEM_Index(mem, 1)
        L.dsp_STORE(8)
        L.dsp_LOAD_GAIN_Fixed(IN + 1, 0.5)
        L.dsp_LOAD_MEM_Index(mem, 1)                         # written earlier in the frame: this frame's value
        L.dsp_ADDXY(); L.dsp_SAT0DB(); L.dsp_STORE(9)
        L.addCode((DSP_LOAD_MEM_DATA << 16) | 2); L.addCode(mux_result)
        L.dsp_STORE(10)
        L.dsp_LOAD(IN + 2); L.dsp_FIR(fdel); L.dsp_STORE(11)
        for k, n in enumerate(sorted(firs)):
            L.dsp_LOAD_GAIN_Fixed(IN + (k % 8), 0.5); L.dsp_FIR(firs[n]); L.dsp_SAT0DB_TPDF_GAIN_Fixed(0.9); L.dsp_STORE(12 + k)
        L.dsp_LOAD(8)                                        # a slot stored earlier in this frame
        L.dsp_GAIN_Fixed(0.5)
        L.dsp_STORE(20)

    return encode(build, fmt)


@pytest.mark.parametrize("fmt", [2, 3, 4, 5, 6])
@pytest.mark.parametrize("block", [1, 3, 64, 65, 130, 700])
def test_frame_parallel_forms(fmt, block):
    prog = forms_program(fmt)
    nframes = 700 if block > 3 else 150
    x = pb.lcg_input(nframes, 8, fmt in (5, 6), seed=3)
    run_both(fmt, prog, x, 24, IN, 0, block, expect_wave=True if block > 1 else False)    # single frames: frame by frame
    # and the frame-by-frame kernel on the same program
    if block in (3, 700):
        run_both(fmt, prog, x, 24, IN, 0, block, expect_wave=False, impl=0)


def carried_programs():
    """name -> (builder, carried?) : what one frame leaves for the next, in every way the opcode set offers"""
    def slot_feedback(L):                # IO 40 is read before it is stored: last frame's value
        L.dsp_CORE()
        L.dsp_LOAD(40); L.dsp_GAIN_Fixed(0.5); L.dsp_COPYXY()
        L.dsp_LOAD_GAIN_Fixed(IN + 0, 0.5); L.dsp_ADDXY(); L.dsp_SAT0DB(); L.dsp_STORE(40); L.dsp_STORE(0)

    def slot_feed_forward(L):            # stored, then loaded: this frame's value
        L.dsp_CORE()
        L.dsp_LOAD_GAIN_Fixed(IN + 0, 0.5); L.dsp_STORE(40); L.dsp_STORE(1)
        L.dsp_LOAD(40); L.dsp_GAIN_Fixed(0.5); L.dsp_STORE(0)

    def mem_feedback(L):                 # LOAD_MEM before the STORE_MEM of the same word
        L.dsp_PARAM(); m = L.dspMem_Location()
        L.dsp_CORE()
        L.dsp_LOAD_GAIN_Fixed(IN + 1, 0.5); L.dsp_COPYXY(); L.dsp_LOAD_MEM(m); L.dsp_GAIN_Fixed(0.5); L.dsp_ADDXY()
        L.dsp_SAT0DB(); L.dsp_STORE_MEM(m); L.dsp_STORE(0)

    def mem_feed_forward(L):
        L.dsp_PARAM(); m = L.dspMem_Location()
        L.dsp_CORE()
        L.dsp_LOAD_GAIN_Fixed(IN + 1, 0.5); L.dsp_STORE_MEM(m); L.dsp_STORE(1)
        L.dsp_LOAD_GAIN_Fixed(IN + 2, 0.5); L.dsp_COPYXY(); L.dsp_LOAD_MEM(m); L.dsp_ADDXY(); L.dsp_SAT0DB(); L.dsp_STORE(0)

    def mem_read_then_overwritten(L):    # reads this frame's value, then a later strand overwrites: still this frame's
        L.dsp_PARAM(); m = L.dspMem_Location()
        L.dsp_CORE()
        L.dsp_LOAD_GAIN_Fixed(IN + 1, 0.5); L.dsp_STORE_MEM(m)
        L.dsp_LOAD_MEM(m); L.dsp_STORE(0)
        L.dsp_LOAD_GAIN_Fixed(IN + 2, 0.25); L.dsp_STORE_MEM(m)
        L.dsp_LOAD_MEM(m); L.dsp_STORE(1)

    def gain_computed_on_the_fly(L):     # STORE_MEM into a gain word that a later GAIN of the same frame reads
        L.dsp_PARAM(); g = L.dspGain_Default(0.25); L.dspGain_Default(0.0)       # (room for a 2-word accumulator)
        L.dsp_CORE()
        L.dsp_LOAD_GAIN_Fixed(IN + 1, 0.5); L.dsp_STORE_MEM(g); L.dsp_STORE(1)
        L.dsp_LOAD(IN + 2); L.dsp_GAIN(g); L.dsp_SAT0DB(); L.dsp_STORE(0)

    def tpdf_calc_late(L):               # the dither value a SAT0DB_TPDF in front of TPDF_CALC sees is last frame's
        L.dsp_CORE()
        L.dsp_LOAD_GAIN_Fixed(IN + 0, 0.5); L.dsp_SAT0DB_TPDF(); L.dsp_STORE(0)
        L.dsp_TPDF_CALC(16)
        L.dsp_LOAD_GAIN_Fixed(IN + 1, 0.5); L.dsp_SAT0DB_TPDF(); L.dsp_STORE(1)

    def tpdf_calc_first_other_width(L):  # first frame ever: no draw (dsp_tpdf.h:55-80), then one per frame
        L.dsp_CORE()
        t = L.dsp_TPDF_CALC(16)
        L.dsp_STORE(2)
        L.addCode((DSP_LOAD_MEM_DATA << 16) | 2); L.addCode(t); L.dsp_STORE(3)
        L.dsp_LOAD_GAIN_Fixed(IN + 1, 0.5); L.dsp_SAT0DB_TPDF(); L.dsp_STORE(1)
        L.dsp_WHITE(); L.dsp_STORE(0)

    def two_cores_through_the_frame(L):  # core 2 reads what core 1 stored: block by block that is the last frame of
        L.dsp_CORE()                     # core 1's block for every frame of core 2's (a constant for core 2)
        L.dsp_LOAD_GAIN_Fixed(IN + 0, 0.5); L.dsp_DELAY_1(); L.dsp_STORE(41); L.dsp_STORE(0)
        L.dsp_CORE()
        L.dsp_LOAD(41); L.dsp_GAIN_Fixed(0.5); L.dsp_STORE(1)

    return {
        "slot_feedback": (slot_feedback, True), "slot_feed_forward": (slot_feed_forward, False),
        "mem_feedback": (mem_feedback, True), "mem_feed_forward": (mem_feed_forward, False),
        "mem_read_then_overwritten": (mem_read_then_overwritten, False),
        "gain_computed_on_the_fly": (gain_computed_on_the_fly, True),
        "tpdf_calc_late": (tpdf_calc_late, True), "tpdf_calc_first_other_width": (tpdf_calc_first_other_width, False),
        "two_cores_through_the_frame": (two_cores_through_the_frame, False),
    }


@pytest.mark.parametrize("name", sorted(carried_programs()))
@pytest.mark.parametrize("fmt", [2, 3, 6])
def test_carried_values_select_the_frame_by_frame_kernel(name, fmt):
    build, carried = carried_programs()[name]
    if name == "gain_computed_on_the_fly" and fmt == 6:
        pytest.skip("the low word of a double is no gain")
    prog = encode(build, fmt)
    x = pb.lcg_input(200, 8, fmt == 6, seed=9)
    for block in (2, 50, 200):
        run_both(fmt, prog, x, 24, IN, 0, block, expect_wave=not carried)


@pytest.mark.parametrize("fmt", [2, 6])
def test_windows_decide_per_call(fmt):
    """A slot the core reads before storing it is the caller's when the call's windows contain it (each frame
    starts from the caller's rows) and the frame's own otherwise: the same program takes either kernel."""
    build, _ = carried_programs()["slot_feedback"]
    prog = encode(build, fmt)
    x = pb.lcg_input(200, 8, fmt == 6, seed=9)
    run_both(fmt, prog, x, 24, IN, 0, 200, expect_wave=False)          # IO 40 outside [0,24) and [32,40)
    run_both(fmt, prog, x, 41, IN, 0, 200, expect_wave=True)           # output window [0,41) holds IO 40


def test_reference_programs_take_the_frame_parallel_kernel():
    """The reference's own example programs (crossoverLV6, dacdiy1) carry nothing between frames."""
    import os
    from tests.golden_recipes import GOLDEN_DIR
    for name, in_stride, in_base, out_stride in (("crossoverLV6.bin", 16, 8, 32), ("dacdiy1.bin", 16, 8, 32)):
        prog = np.fromfile(os.path.join(GOLDEN_DIR, name), dtype=np.uint32)
        x = pb.lcg_input(300, in_stride, False, seed=5)
        run_both(2, prog, x, out_stride, in_base, 0, 300, expect_wave=True, span=64)


def test_switching_kernels_mid_stream_keeps_the_state():
    """dspRuntimeSetOption("interp_impl" / "generic") between blocks only re-lowers the cores: delay lines,
    filter state, dither generator and the persistent frame stay on the device."""
    fmt = 6
    prog = forms_program(fmt)
    x = pb.lcg_input(600, 8, True, seed=4)
    o = po.OracleProgram(fmt, prog, fs=48000, random=11, dither=24)
    frame = np.zeros(4096, dtype=np.uint32)
    want = o.run_block(x, 24, IN, 0, block=100, frame=frame)
    r = rt.Runtime(fmt, prog, fs=48000, random=11, dither=24)
    try:
        got = np.zeros_like(want)
        for k, b0 in enumerate(range(0, 600, 100)):
            r.set_option("interp_impl", k % 2)
            got[b0:b0 + 100] = r.run_block(x[b0:b0 + 100], 24, IN, 0)
        assert (got.view(np.uint32) == want.view(np.uint32)).all()
        assert (r.sync_state() == o.state).all()
    finally:
        r.set_option("interp_impl", 1)
        r.release()


def _all_vs_per_core(fmt, prog, x, out_stride, in_base, out_base, block, fs=48000, seed=11):
    """dspRuntimeBlockAll against the per-core calls (and the oracle): outputs and the whole buffer"""
    o = po.OracleProgram(fmt, prog, fs=fs, random=seed, dither=24)
    frame = np.zeros(4096, dtype=np.uint32)
    want = o.run_block(x, out_stride, in_base, out_base, block=block, frame=frame)
    r = rt.Runtime(fmt, prog, fs=fs, random=seed, dither=24)
    assert r.rc == o.rc and r.rc > 0
    try:
        got = r.run_block_all(x, out_stride, in_base, out_base, block=block)
        levels, cores = r.get_option("levels"), r.get_option("cores")
        bad = np.nonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=0))[0]
        assert bad.size == 0, f"DSP_FORMAT {fmt} block {block}: output columns {list(bad)} differ ({cores} cores in {levels} levels)"
        r.sync_state()
        n = int(prog[1]) + int(prog[2])
        words = np.nonzero(r.buf[12:n] != o.buf[12:n])[0] + 12
        assert words.size == 0, f"DSP_FORMAT {fmt} block {block}: buffer words {list(words[:8])} differ ({cores} cores in {levels} levels)"
        return levels, cores
    finally:
        r.release()


def test_block_all_on_the_reference_programs():
    """crossoverLV6: core 2 dithers with core 1's TPDF value -> two levels.  dacdiy1: core 1 feeds cores 2-4 through
    STORE_MEM / LOAD_MEM and the dither value, those three do not meet -> two levels for four cores."""
    import os
    from tests.golden_recipes import GOLDEN_DIR
    for name, expect in (("crossoverLV6.bin", (2, 2)), ("dacdiy1.bin", (2, 4))):
        prog = np.fromfile(os.path.join(GOLDEN_DIR, name), dtype=np.uint32)
        x = pb.lcg_input(700, 16, False, seed=5)
        for block in (1, 100, 700):
            assert _all_vs_per_core(2, prog, x, 32, 8, 0, block) == expect


@pytest.mark.parametrize("seed", range(100, 130))
def test_block_all_on_random_programs(seed):
    """Random multi-core programs (memories, shared dither, histogram slots ...): whatever levels the host finds,
    the result is the per-core one."""
    from tests.fuzz_programs import IN_BASE, N_IN, N_OUT, random_program
    for fmt in (2, 5, 6):
        prog = random_program(seed, fmt)
        fs, block = [48000, 48000, 96000][seed % 3], [7, 64, 400][seed % 3]
        x = pb.lcg_input(400, N_IN, fmt in (5, 6), seed=seed)
        _all_vs_per_core(fmt, prog, x, N_OUT, IN_BASE, 0, block, fs=fs, seed=seed)


def test_block_all_runs_independent_cores_side_by_side():
    """four cores that share nothing -> one level; outputs interleaved in the same rows"""
    def build(L):
        L.dsp_PARAM()
        banks = []
        for c in range(4):
            b = L.dspBiquad_Sections(3)
            for k in range(3):
                L.dsp_Filter2ndOrder(FPEAK, 200.0 * (k + 1) + 50 * c, 1.1, 0.9)
            banks.append(b)
        for c in range(4):
            L.dsp_CORE()
            L.dsp_LOAD_GAIN_Fixed(IN + c, 0.5); L.dsp_DELAY_1(); L.dsp_BIQUADS(banks[c]); L.dsp_DCBLOCK(10); L.dsp_SAT0DB()
            L.dsp_STORE(c); L.dsp_STORE(8 + c)
    for fmt in (2, 6):
        prog = encode(build, fmt)
        x = pb.lcg_input(500, 8, fmt == 6, seed=2)
        for block in (64, 500):
            assert _all_vs_per_core(fmt, prog, x, 16, IN, 0, block) == (1, 4)


def test_block_all_mixes_chain_cores_and_interpreted_cores():
    """a chain core (parallel kernels) next to interpreted cores: the chain core keeps its place in the order"""
    def build(L):
        L.dsp_PARAM()
        b = L.dspBiquad_Sections(2)
        for k in range(2):
            L.dsp_Filter2ndOrder(FPEAK, 500.0 * (k + 1), 1.0, 0.9)
        L.dsp_CORE()                                                             # chain core
        L.dsp_LOAD_GAIN_Fixed(IN + 0, 0.5); L.dsp_BIQUADS(b); L.dsp_SAT0DB(); L.dsp_STORE(0)
        L.dsp_CORE()                                                             # interpreted
        L.dsp_LOAD_GAIN_Fixed(IN + 1, 0.5); L.dsp_DELAY_1(); L.dsp_STORE(1)
        L.dsp_CORE()                                                             # interpreted, reads what the chain core stored
        L.dsp_LOAD(0); L.dsp_GAIN_Fixed(0.5); L.dsp_DELAY_1(); L.dsp_STORE(2)
    for fmt in (2, 6):
        prog = encode(build, fmt)
        x = pb.lcg_input(300, 8, fmt == 6, seed=6)
        r = rt.Runtime(fmt, prog)
        assert r.core_info(0)["chains"] == 1 and r.core_info(1)["chains"] == 0
        r.release()
        for block in (50, 300):
            levels, cores = _all_vs_per_core(fmt, prog, x, 8, IN, 0, block, seed=0)
            assert cores == 3 and levels >= 2


def test_block_all_device_entry_point_on_a_side_stream():
    torch = pytest.importorskip("torch")
    import os
    from tests.golden_recipes import GOLDEN_DIR
    prog = np.fromfile(os.path.join(GOLDEN_DIR, "dacdiy1.bin"), dtype=np.uint32)
    xh = pb.lcg_input(512, 16, False, seed=5)
    o = po.OracleProgram(2, prog, fs=48000, random=3, dither=24)
    want = o.run_block(xh, 8, 8, 0, block=256, frame=np.zeros(4096, dtype=np.uint32))
    r = rt.Runtime(2, prog, fs=48000, random=3, dither=24)
    try:
        r.set_option("profile", 1)
        side = torch.cuda.Stream()
        x = dm.to_device(xh)
        y = torch.zeros((512, 8), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for b0 in (0, 256):
                r._check(r.L.dspRuntimeBlockAllDevice(2, r.rundata, x[b0:].data_ptr(), 16, 8, y[b0:].data_ptr(), 8, 0, 256, side.cuda_stream))
        side.synchronize()
        assert (dm.to_host(y) == want).all()
        assert (r.sync_state() == o.state).all()
        assert r.get_option("levels") == 2 and r.get_option("cores") == 4
        assert r.kernel_time(KIND_WAVE)[1] >= 4                # two blocks x two levels, the cores of a level in one launch
    finally:
        r.set_option("profile", 0)
        r.release()


def test_block_all_orders_a_core_behind_the_one_that_computes_its_gain():
    """core 1 stores a gain with STORE_MEM, core 2 applies it with DSP_GAIN: a parameter read, not a LOAD_MEM"""
    def build(L):
        L.dsp_PARAM(); g = L.dspGain_Default(0.25); L.dspGain_Default(0.0)
        L.dsp_CORE()
        L.dsp_LOAD_GAIN_Fixed(IN + 1, 0.5); L.dsp_DELAY_1(); L.dsp_STORE_MEM(g); L.dsp_STORE(1)
        L.dsp_CORE()
        L.dsp_LOAD(IN + 2); L.dsp_GAIN(g); L.dsp_DELAY_1(); L.dsp_SAT0DB(); L.dsp_STORE(0)
        L.dsp_CORE()
        L.dsp_LOAD(IN + 3); L.dsp_DELAY_1(); L.dsp_STORE(2)
    for fmt in (2, 3):
        prog = encode(build, fmt)
        x = pb.lcg_input(300, 8, False, seed=8)
        for block in (7, 300):
            assert _all_vs_per_core(fmt, prog, x, 8, IN, 0, block, seed=0) == (2, 3)


@pytest.mark.parametrize("fmt", [3, 4, 5, 6])
def test_long_fir_in_an_interpreted_core(fmt):
    """a 3000-tap DSP_FIR next to a delay line (so the core is not a chain): the frame-parallel FIR reads a
    3000 + 64 word sequence from LDS, every lane in the reference's tap order"""
    taps = (np.random.default_rng(3).uniform(-1, 1, 3000) / 200).astype(np.float32)

    def build(L):
        L.dsp_PARAM()
        imp = L.dspFir_Impulses()
        L.dspFir_ImpulseData(taps.ctypes.data_as(C.POINTER(C.c_float)), 3000)
        L.dsp_CORE()
        L.dsp_LOAD_GAIN_Fixed(IN, 0.9); L.dsp_DELAY_1(); L.dsp_FIR(imp); L.dsp_SAT0DB(); L.dsp_STORE(0)
    prog = encode(build, fmt)
    x = pb.lcg_input(500, 8, fmt in (5, 6), seed=12)
    for block in (64, 500):
        run_both(fmt, prog, x, 8, IN, 0, block, expect_wave=True)
    run_both(fmt, prog, x, 8, IN, 0, 500, expect_wave=False, impl=0)


def test_io_numbers_beyond_the_frame_parallel_limit():
    """IO numbers >= 256: more frame than the per-lane layout holds -> frame by frame, the core runs alone"""
    def build(L):
        L.dsp_CORE()
        L.dsp_LOAD_GAIN_Fixed(300, 0.5); L.dsp_DELAY_1(); L.dsp_STORE(2)
        L.dsp_CORE()
        L.dsp_LOAD_GAIN_Fixed(301, 0.25); L.dsp_DELAY_1(); L.dsp_STORE(3)
    for fmt in (2, 6):
        prog = encode(build, fmt, max_io=320)
        x = pb.lcg_input(200, 8, fmt == 6, seed=4)
        run_both(fmt, prog, x, 8, 298, 0, 200, expect_wave=False, span=320)
        assert _all_vs_per_core(fmt, prog, x, 8, 298, 0, 100, seed=0) == (2, 2)


def test_wide_core_is_cut_into_strand_groups():
    """100 channel strands in ONE core (gain, cascade, delay line each -- not a chain core): dspRuntimeBlockAll cuts
    it into groups of strands that do not depend on each other and runs them as one grid; same result as the
    core in one piece"""
    nch = 100

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
    for fmt in (2, 6):
        prog = encode(build, fmt, max_io=256)
        x = pb.lcg_input(256, nch, fmt == 6, seed=21)
        o = po.OracleProgram(fmt, prog, fs=48000, random=5, dither=24)
        want = o.run_block(x, nch, 128, 0, block=128, frame=np.zeros(4096, dtype=np.uint32))
        for split in (2, 1, 0):                 # 2: the run of strands on lanes (the default), 1: strand groups of the interpreter, 0: whole
            r = rt.Runtime(fmt, prog, fs=48000, random=5, dither=24)
            r.set_option("strand_split", 1 if split else 0)
            r.set_option("strand_lanes", 1 if split == 2 else 0)
            try:
                got = r.run_block_all(x, nch, 128, 0, block=128)
                assert (got.view(np.uint32) == want.view(np.uint32)).all(), (fmt, split)
                assert (r.sync_state() == o.state).all(), (fmt, split)
                pieces, levels = r.get_option("pieces"), r.get_option("levels")
                if split == 2:
                    assert pieces == 2 and levels == 2 and r.get_option("strands") == nch, (pieces, levels)   # the TPDF_CALC, then the run
                elif split:
                    assert pieces > 20 and levels == 2, (pieces, levels)       # TPDF_CALC's piece first, the rest together
                else:
                    assert pieces == 1 and levels == 1
            finally:
                r.set_option("strand_split", 1)
                r.set_option("strand_lanes", 1)
                r.release()


def test_strands_that_hand_values_over_stay_in_one_piece():
    """a strand that loads a slot or a memory an earlier strand of the same core stored must see THIS frame's value:
    such strands are never separated (the pieces of one core may not meet at all)"""
    def build(L):
        L.dsp_PARAM(); m = L.dspMem_Location()
        L.dsp_CORE()
        L.dsp_LOAD_GAIN_Fixed(IN + 0, 0.5); L.dsp_DELAY_1(); L.dsp_STORE(40); L.dsp_STORE(0)         # IO 40: outside both windows
        L.dsp_LOAD_GAIN_Fixed(IN + 1, 0.5); L.dsp_DELAY_1(); L.dsp_STORE_MEM(m); L.dsp_STORE(1)
        L.dsp_LOAD(40); L.dsp_GAIN_Fixed(0.5); L.dsp_DELAY_1(); L.dsp_STORE(2)                        # needs strand 1
        L.dsp_LOAD_MEM(m); L.dsp_GAIN_Fixed(0.5); L.dsp_DELAY_1(); L.dsp_STORE(3)                     # needs strand 2
        L.dsp_LOAD_GAIN_Fixed(IN + 2, 0.5); L.dsp_DELAY_1(); L.dsp_DCBLOCK(10); L.dsp_STORE(4)        # needs nobody
        L.dsp_LOAD_GAIN_Fixed(IN + 3, 0.5); L.dsp_COPYXY(); L.dsp_LOAD_GAIN_Fixed(IN + 4, 0.5)        # Y handed to the next load's strand
        L.dsp_ADDXY(); L.dsp_SAT0DB(); L.dsp_STORE(5)
    for fmt in (2, 6):
        prog = encode(build, fmt)
        x = pb.lcg_input(300, 8, fmt == 6, seed=13)
        for block in (2, 64, 300):
            levels, cores = _all_vs_per_core(fmt, prog, x, 8, IN, 0, block, seed=0)
            assert cores == 1


@pytest.mark.parametrize("seed,switch", [(s, "WIDE") for s in list(range(1030, 1042)) + [37, 211]] +
                                        [(s, "NAN_HEAVY") for s in (10021, 7003, 7011, 12005)])
def test_block_all_on_wide_and_nan_heavy_random_programs(seed, switch, monkeypatch):
    """the generator's development switches: many strands per core (strand groups, joins) and DITHER-poisoned
    strands (NaN operand order: seeds 1037 and 10021 found AVGYX and DCBLOCK)"""
    import tests.fuzz_programs as fz
    monkeypatch.setattr(fz, switch, True)
    for fmt in (2, 3, 6):
        prog = fz.random_program(seed, fmt)
        fs, block = [48000, 48000, 96000][seed % 3], [5, 64, 150][seed % 3]
        x = pb.lcg_input(150, fz.N_IN, fmt in (5, 6), seed=seed)
        _all_vs_per_core(fmt, prog, x, fz.N_OUT, fz.IN_BASE, 0, block, fs=fs, seed=seed)


@pytest.mark.parametrize("fmt", [2, 3, 4, 5, 6])
def test_two_way_strands_fork_into_two_pieces(fmt):
    """LOAD, COPYXY, <way 1>, SWAPXY, <way 2> (crossoverLV6.bin's second core): way 2 only needs the load's value, so it runs as a
    piece of its own -- the load again, then what follows the SWAPXY -- beside way 1.  The subtractive shape (way 2 takes Y - X)
    must stay in one piece.  Both against the oracle, outputs and state, blocks around the batch size."""
    from tests.test_gpu_strands import crossover_program
    for shape, nch, pieces in (("two_way", 3, 1 + 2 * 3), ("subtractive", 3, 1 + 3)):
        prog = crossover_program(nch, fmt, shape)
        x = pb.lcg_input(300, nch, fmt in (5, 6), seed=31)
        o = po.OracleProgram(fmt, prog, fs=48000, random=3, dither=24)
        blocks = [1, 64, 100, 135]
        want = np.concatenate([o.run_block(x[a:a + n], 2 * nch, 128) for a, n in zip(np.cumsum([0] + blocks[:-1]), blocks)])
        r = rt.Runtime(fmt, prog, fs=48000, random=3, dither=24)
        try:
            got = np.concatenate([r.run_block_all(x[a:a + n], 2 * nch, 128) for a, n in zip(np.cumsum([0] + blocks[:-1]), blocks)])
            assert r.get_option("pieces") == pieces, (shape, r.get_option("pieces"))
            assert (got.view(np.uint32) == want.view(np.uint32)).all(), (fmt, shape)
            assert (r.sync_state() == o.state).all(), (fmt, shape)
        finally:
            r.release()
