"""Parity of the HIP path (through the C-ABI library) against
  (1) the golden vectors the compiled reference produced (tests/golden/), and
  (2) the oracle (oracle/liboracle.so) on the same seeded inputs at other shapes.

Bar: bit-exact for DSP_FORMAT 2 (int64 fixed point) outputs AND state; for the float models
(4, 6) the north-star tolerance is 1e-6 relative (to the block's peak magnitude) -- the tests assert
that tolerance everywhere and, where the kernel performs the reference's operations in the
reference's order (biquad cascade, plain FIR, and the MFMA FIR whose K index ascends with the tap
index), additionally assert bit equality."""
import hashlib
import json
import os

import numpy as np
import pytest

from avdsp_amd import progbuilder as pb
from avdsp_amd import runtime as rt
from oracle import pyoracle as po
from tests.golden_recipes import GOLDEN_DIR, check_against_golden, make_input, make_program

pytestmark = pytest.mark.gpu

REL_TOL = 1e-6          # north_star: "within 1e-6 relative in its float mode"

with open(os.path.join(GOLDEN_DIR, "manifest.json")) as _f:
    MANIFEST = json.load(_f)

# chain programs: formats 2, 4, 6 on the pipelined cascade + MFMA FIR, formats 3 and 5 on chain_lane (one lane per chain)
DEVICE_CASES = [c for c in MANIFEST["cases"] if c["program"]["kind"] == "synth"]
# the general device interpreter: every committed / reference-encoded program (none of them is a pure
# set of chains), the random programs of tests/fuzz_programs.py in all five arithmetic models, and the
# synthetic chain programs forced through it (formats 3 and 5 have no other path);
# the long-FIR cases would take minutes on one lane
GENERAL_CASES = [c for c in MANIFEST["cases"] if c["program"]["kind"] in ("file", "fuzz") or
                 (c["nframes"] * max(c["program"].get("taps", 0), 1) * c["program"].get("channels", 1) <= 2_000_000)]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def assert_close(got, want, fmt, exact=True, what="output"):
    g, w = got.view(np.uint32), want.view(np.uint32)
    if fmt == 2:
        assert (g == w).all(), f"{what}: int64 mode must be bit-exact ({np.count_nonzero(g != w)} words differ)"
        return
    if fmt == 4:
        # int32 samples out of a double accumulator: 1 LSB of s.31 is far below 1e-6 relative
        d = np.abs(got.astype(np.int64) - want.astype(np.int64)).max()
        peak = max(np.abs(want.astype(np.int64)).max(), 1)
        assert d <= REL_TOL * peak, f"{what}: max abs diff {d} vs peak {peak}"
    else:
        diff = np.abs(got.astype(np.float64) - want.astype(np.float64))
        d = diff.max()
        peak = max(np.abs(want.astype(np.float64)).max(), 1e-30)
        bad_cols = np.nonzero(diff.max(axis=0) > REL_TOL * peak)[0] if diff.ndim == 2 else []
        assert d <= REL_TOL * peak, f"{what}: max abs diff {d} vs peak {peak}; columns {list(bad_cols)[:8]}, first frame {int(np.argmax(diff.max(axis=1) > REL_TOL * peak)) if diff.ndim == 2 else -1}"
    if exact:
        assert (g == w).all(), f"{what}: expected bit-exact, {np.count_nonzero(g != w)} words differ"


@pytest.fixture(autouse=True)
def _release():
    yield
    rt.lib().dspRuntimeRelease()


@pytest.mark.parametrize("case", DEVICE_CASES, ids=lambda c: c["name"])
@pytest.mark.parametrize("impl", ["default", "simple_plain"])
def test_golden(case, impl):
    fmt = case["fmt"]
    prog = make_program(case["program"])
    x = make_input(case["input"], fmt)
    r = rt.Runtime(fmt, prog, fs=case["fs"], random=case["random"], dither=case["dither"])
    assert r.rc == case["init_rc"]
    if impl == "simple_plain":
        r.set_option("biquad_impl", 0)
        r.set_option("fir_impl", 0)
    else:
        r.set_option("biquad_impl", 1)
        r.set_option("fir_impl", 1)
    out = r.run_block(x, case["out_stride"], case["in_base"], case["out_base"], block=case["block"])
    g = np.load(os.path.join(GOLDEN_DIR, case["name"] + ".npz"))
    assert_close(out[:16], g["head"], fmt, what="head")
    assert_close(out[-16:], g["tail"], fmt, what="tail")
    state = r.sync_state()
    if case["full"]:
        assert_close(out, g["out"], fmt)
        assert (state == g["state"]).all(), "state area differs from the reference's"
    assert sha(out) == case["out_sha"]
    assert sha(state) == case["state_sha"]


with open(os.path.join(GOLDEN_DIR, "hilbert_manifest.json")) as _f:
    HILBERT = json.load(_f)


@pytest.mark.parametrize("generic", [0, 1], ids=["chain_kernels", "interpreter"])
@pytest.mark.parametrize("case", HILBERT, ids=lambda c: c["name"])
def test_hilbert_programs_match_the_reference(case, generic):
    """Programs made of dsp_Hilbert banks by the REFERENCE encoder, run by the reference runtime (tests/golden/make_hilbert_goldens.py):
    one core of 22 chains with 1 .. 10 all-pass cells each, two chains per input -- on the chain kernels (cascades of ten different
    lengths in one launch group set; formats 3 / 5: chain_rows) and on the interpreter, outputs and state bit for bit."""
    fmt = case["fmt"]
    prog = make_program(case["program"])
    x = make_input(case["input"], fmt)
    r = rt.Runtime(fmt, prog, fs=case["fs"], random=case["random"], dither=case["dither"])
    assert r.rc == case["init_rc"]
    r.set_option("generic", generic)
    try:
        out = r.run_block(x, case["out_stride"], case["in_base"], case["out_base"], block=case["block"])
        assert (r.core_info(0)["chains"] == 0) == bool(generic)
        check_against_golden(case, out, r.sync_state(), sha)
    finally:
        r.set_option("generic", 0)


def _oracle_vs_device(fmt, prog, x, out_stride, in_base, blocks, fs=48000, dither=31, opts=None, exact=True):
    o = po.OracleProgram(fmt, prog, fs=fs, dither=dither)
    r = rt.Runtime(fmt, prog, fs=fs, dither=dither)
    assert r.rc == o.rc
    for k, v in (opts or {}).items():
        r.set_option(k, v)
    pos = 0
    for b in blocks:
        xo = x[pos:pos + b]
        want = o.run_block(xo, out_stride, in_base)
        got = r.run_block(xo, out_stride, in_base)
        assert_close(got, want, fmt, exact=exact, what=f"block at frame {pos}")
        pos += b
    assert (r.sync_state() == o.state).all(), "state differs after the last block"
    return r


@pytest.mark.parametrize("fmt", [2, 4, 6])
@pytest.mark.parametrize("channels,sections", [(1, 1), (3, 2), (8, 8), (37, 5), (64, 16), (5, 16), (2, 24), (2, 40)])
def test_biquad_vs_oracle_ragged_blocks(fmt, channels, sections):
    """State must carry across blocks of uneven size (1-frame, odd, > prefetch depth)."""
    prog = pb.synth_program(fmt, channels, sections)
    blocks = [1, 7, 16, 100, 33, 1, 2]
    x = pb.lcg_input(sum(blocks), channels, fmt == 6, seed=99 + channels)
    _oracle_vs_device(fmt, prog, x, channels, channels, blocks)


@pytest.mark.parametrize("fmt", [4, 6])
@pytest.mark.parametrize("taps", [1, 2, 7, 64, 255, 300, 1025])
@pytest.mark.parametrize("fir_impl", [0, 1])
def test_fir_vs_oracle_ragged_blocks(fmt, taps, fir_impl):
    prog = pb.synth_program(fmt, 3, 0, taps)
    blocks = [1, 5, 256, 257, 40, 1030]
    x = pb.lcg_input(sum(blocks), 3, fmt == 6, seed=5 + taps)
    _oracle_vs_device(fmt, prog, x, 3, 3, blocks, opts={"fir_impl": fir_impl})


@pytest.mark.parametrize("fmt", [4, 6])
def test_mixed_chain_vs_oracle(fmt):
    prog = pb.synth_program(fmt, 6, 4, 129)
    blocks = [300, 1, 64, 700]
    x = pb.lcg_input(sum(blocks), 6, fmt == 6, seed=77)
    _oracle_vs_device(fmt, prog, x, 6, 6, blocks)


@pytest.mark.parametrize("fmt", [2, 4, 6])
def test_saturation_and_fullscale(fmt):
    prog = pb.synth_program(fmt, 8, 8, gain=7.5)
    x = make_input(dict(kind="fullscale", frames=300, channels=8), fmt)
    _oracle_vs_device(fmt, prog, x, 8, 8, [300])


@pytest.mark.parametrize("fmt", [2, 4, 6])
def test_single_frame_entry_point(fmt):
    """dspRuntime_N(core, rundata, samples): one frame, samples[] indexed by IO number, in place."""
    C, S = 4, 3
    prog = pb.synth_program(fmt, C, S)
    o = po.OracleProgram(fmt, prog)
    r = rt.Runtime(fmt, prog)
    x = pb.lcg_input(12, C, fmt == 6, seed=3)
    for n in range(12):
        frame_o = np.zeros(2 * C, dtype=rt.sample_dtype(fmt)); frame_o[C:] = x[n]
        frame_d = frame_o.copy()
        po.lib().oracle_run(o.ctx, o.cores[0], o.data_ptr, frame_o.ctypes.data)
        assert r.run_frame(frame_d) == 0
        assert_close(frame_d, frame_o, fmt, what=f"frame {n}")
    assert (r.sync_state() == o.state).all()


@pytest.mark.parametrize("fmt", [2, 6])
def test_checkpoint_restore(fmt):
    """The caller's buffer is the checkpoint: sync, copy, reload elsewhere, continue -> same stream."""
    prog = pb.synth_program(fmt, 5, 4, 0 if fmt == 2 else 33)
    x = pb.lcg_input(400, 5, fmt == 6, seed=11)
    r = rt.Runtime(fmt, prog)
    first = r.run_block(x[:150], 5, 5)
    saved = r.sync_state().copy()
    rest = r.run_block(x[150:], 5, 5)
    r.release()
    r2 = rt.Runtime(fmt, prog)
    r2.state[:] = saved
    r2.upload_state()
    rest2 = r2.run_block(x[150:], 5, 5)
    assert (rest.view(np.uint32) == rest2.view(np.uint32)).all()
    assert first.shape == (150, 5)


def test_bypass_and_multi_bank_and_plain_load():
    """LOAD (no gain), two banks in one chain (second bypassed), two STOREs of one value."""
    fmt = 6
    pw = pb.ProgramWriter(fmt)
    pw.core()
    pw.param()
    b1 = pw.biquad_bank(pb.synth_sections(0, 2, pb.F48000, pb.F48000))
    pw.param()
    b2 = pw.biquad_bank(pb.synth_sections(1, 3, pb.F48000, pb.F48000), bypass=0)
    pw.param()
    b3 = pw.biquad_bank(pb.synth_sections(2, 1, pb.F48000, pb.F48000))
    pw.load(4)
    pw.biquads(b1, 2)
    pw.biquads(b2, 3)
    pw.biquads(b3, 1)
    pw.sat0db()
    pw.store(0)
    pw.store(2)
    pw.load_gain_fixed(5, 0.5)
    pw.store(1)
    prog = pw.end_of_code()
    x = pb.lcg_input(200, 2, True, seed=8)
    o = po.OracleProgram(fmt, prog)
    r = rt.Runtime(fmt, prog)
    want = o.run_block(x, 4, 4)
    got = r.run_block(x, 4, 4)
    assert_close(got, want, fmt)
    assert (got[:, 3] == 0).all()           # IO 3 is never stored: untouched
    assert (r.sync_state() == o.state).all()
    assert r.core_info() == dict(chains=2, max_sections=3, max_taps=0)


@pytest.mark.parametrize("case", GENERAL_CASES, ids=lambda c: c["name"])
def test_golden_general_interpreter(case):
    """Programs with X/Y arithmetic, dither, delay lines, mixers, meters ... against
    the vectors the compiled reference produced: outputs and the final state area, bit for bit, in
    all five arithmetic models."""
    fmt = case["fmt"]
    prog = make_program(case["program"])
    x = make_input(case["input"], fmt)
    r = rt.Runtime(fmt, prog, fs=case["fs"], random=case["random"], dither=case["dither"])
    assert r.rc == case["init_rc"]
    if case["program"]["kind"] == "synth":
        r.set_option("generic", 1)
    try:
        out = r.run_block(x, case["out_stride"], case["in_base"], case["out_base"], block=case["block"])
        if r.cores:
            assert r.core_info(0)["chains"] == 0            # it really was the interpreter
        check_against_golden(case, out, r.sync_state(), sha)
    finally:
        r.set_option("generic", 0)


@pytest.mark.parametrize("interp_impl", [1, 0], ids=["frame_parallel", "frame_by_frame"])
@pytest.mark.parametrize("seed", list(range(12, 72)) + [344, 373, 416, 462, 2260, 6273])     # the last six: NaN payloads through float / double adds
def test_random_programs_vs_oracle(seed, interp_impl):
    """More random programs than there are goldens: the interpreter against the oracle (itself held to the
    compiled reference on these generators, tests/golden/make_goldens.py and 800+ runs while developing),
    all five arithmetic models, outputs and the whole buffer (state, STORE_MEM targets) bit for bit; with
    the frame-parallel kernel wherever the host finds a core eligible, and frame by frame throughout."""
    from tests.fuzz_programs import IN_BASE, N_IN, N_OUT, random_program
    rt.Runtime.set_global_option("interp_impl", interp_impl)
    try:
        _random_programs(seed, IN_BASE, N_IN, N_OUT, random_program)
    finally:
        rt.Runtime.set_global_option("interp_impl", 1)


def _random_programs(seed, IN_BASE, N_IN, N_OUT, random_program):
    for fmt in (2, 3, 4, 5, 6):
        prog = random_program(seed, fmt)
        fs, block = [48000, 48000, 96000][seed % 3], [1, 64, 500][seed % 3]
        x = pb.lcg_input(500, N_IN, fmt in (5, 6), seed=seed)
        o = po.OracleProgram(fmt, prog, fs=fs, random=seed, dither=24)
        r = rt.Runtime(fmt, prog, fs=fs, random=seed, dither=24)
        assert r.rc == o.rc
        if r.rc < 0:
            continue
        want = o.run_block(x, N_OUT, IN_BASE, 0, scratch_len=48, block=block)
        got = r.run_block(x, N_OUT, IN_BASE, 0, block=block)
        bad = np.nonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=0))[0]
        assert bad.size == 0, f"seed {seed} DSP_FORMAT {fmt}: output columns {list(bad)} differ"
        r.sync_state()
        n = int(prog[1]) + int(prog[2])
        assert (r.buf[12:n] == o.buf[12:n]).all(), f"seed {seed} DSP_FORMAT {fmt}: buffer differs after the run"
        r.release()


@pytest.mark.parametrize("seed", range(40))
def test_random_chains_with_stress_inputs(seed):
    """Random chain programs (channels, sections, taps, rate column, gain, dither, block size) fed zeros and
    negative zeros, full scale and beyond, barely-normal and subnormal samples: the parallel kernels
    (formats 2, 4, 6; both implementations) and the interpreter (3, 5) against the oracle, which agrees
    with the compiled reference on this generator (300 runs while developing)."""
    from tests.fuzz_programs import random_chain_case, stress_input
    rng, C, S, T, fmin, fmax, gain, fs, n, dither = random_chain_case(seed)
    for fmt in (2, 3, 4, 5, 6):
        taps = 0 if fmt == 2 else T
        if S == 0 and taps == 0:
            continue
        prog = pb.synth_program(2 if fmt == 2 else 6, C, S, taps, fmin, fmax, gain)
        x = stress_input(rng, n, C, fmt in (5, 6))
        block = int(rng.choice([1, 7, 64, n])) if n < 120 else int(rng.choice([7, 64, n]))
        for impl in ((1, 1), (0, 0)) if fmt in (2, 4, 6) else ((1, 1),):
            o = po.OracleProgram(fmt, prog, fs=fs, dither=dither)
            r = rt.Runtime(fmt, prog, fs=fs, dither=dither)
            r.set_option("biquad_impl", impl[0]); r.set_option("fir_impl", impl[1])
            want = o.run_block(x, C, C, 0, block=block)
            got = r.run_block(x, C, C, 0, block=block)
            assert (got.view(np.uint32) == want.view(np.uint32)).all(), f"seed {seed} DSP_FORMAT {fmt} impl {impl}: outputs differ"
            assert (r.sync_state() == o.state).all(), f"seed {seed} DSP_FORMAT {fmt} impl {impl}: state differs"
            r.release()


def test_general_interpreter_single_frame_and_store_mem():
    """dspRuntime_N on a non-chain program: samples[] in place, frame by frame, equals the block call;
    DSP_STORE_MEM results come back into the program words with dspRuntimeSyncState."""
    prog = np.fromfile(os.path.join(GOLDEN_DIR, "dacdiy1.bin"), dtype=np.uint32)
    x = pb.lcg_input(40, 16, False, seed=5)
    o = po.OracleProgram(2, prog, fs=48000, random=3, dither=24)
    want = o.run_block(x, 32, 8, 0, scratch_len=40, block=1)
    r = rt.Runtime(2, prog, fs=48000, random=3, dither=24)
    got = np.zeros_like(want)
    frame = np.zeros(40, dtype=np.int32)
    for n in range(len(x)):
        frame[8:24] = x[n]
        for core in range(len(r.cores)):
            r.run_frame(frame, core)
        got[n] = frame[:32]
        frame[:32] = 0
    # the oracle's block loop starts every frame from the (zero) output row: same thing
    assert (got == want).all()
    assert (r.sync_state() == o.state).all()
    n = int(prog[1])
    assert (r.buf[:n] == o.buf[:n]).all(), "program words (STORE_MEM targets) differ"
    assert (r.buf[:n] != prog[:n]).any(), "this program is expected to write into its parameter section"


def test_refusals_are_loud():
    """No CPU fallback and no guessing: what neither device path can run fails with a reason."""
    # int64 FIR is undefined behaviour in the reference: refused, not guessed
    prog = pb.synth_program(2, 2, 1, 9)
    r = rt.Runtime(2, prog)
    with pytest.raises(rt.AvdspError) as e:
        r.run_block(np.zeros((4, 2), dtype=np.int32), 2, 2)
    assert e.value.code == -8 and "undefined behaviour" in str(e.value)
    # (a float-encoded program through the int64 entry point is no refusal any more: dspChangeFormat, tests/test_changeformat.py)
    # a state offset outside the data area (the reference would scribble over memory)
    prog = np.fromfile(os.path.join(GOLDEN_DIR, "crossoverLV6.bin"), dtype=np.uint32).copy()
    i = 0
    while (int(prog[i]) >> 16) != 47:                       # DSP_DELAY: [maxSize][data offset][us offset]
        i += int(prog[i]) & 0xFFFF
    prog[i + 2] = 1 << 20
    r = rt.Runtime(2, prog, fs=48000, dither=24)
    assert r.rc == 288                                       # payload words are not part of the checksum
    with pytest.raises(rt.AvdspError) as e:
        r.run_block(np.zeros((4, 16), dtype=np.int32), 32, 8)
    assert e.value.code == -8 and "outside the state area" in str(e.value)


@pytest.mark.parametrize("neg", MANIFEST["init_return_codes"], ids=lambda n: n["case"])
def test_init_return_codes_match_reference(neg):
    good = pb.synth_program(2, 2, 2)
    prog, fs, max_size = good, 48000, None
    c = neg["case"]
    if c == "bad_checksum":
        prog = good.copy(); prog[3] ^= 1
    elif c in ("unsupported_fs", "fs_out_of_range"):
        fs = neg["fs"]
    elif c == "buffer_too_small":
        max_size = neg["max_size"]
    elif c == "no_header":
        prog = good.copy(); prog[0] = (2 << 16) | 12
    elif c == "opcode_too_new":
        prog = good.copy(); prog[6] = (62 << 16) | (int(prog[6]) & 0xFFFF)
    assert rt.Runtime(2, prog, fs=fs, max_size=max_size).rc == neg["rc"]


# ---------------------------------------------------------------------------------------------
# BASELINE.json full sizes: size-independent properties (the oracle would take minutes here)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fmt", [2, 6])
def test_full_size_biquad_reblocking_and_sampled_oracle(fmt):
    """cfg3: 4096 ch x 16 biquads x 1024 frames.  (a) one 1024-frame block == 4 blocks of 256 ==
    ragged blocks, bit for bit (the recurrence has no block-size dependence); (b) 16 sampled
    channels equal the oracle run on those channels alone (channels are independent)."""
    C, S, B = 4096, 16, 1024
    prog = pb.synth_program(fmt, C, S)
    x = pb.lcg_input(B, C, fmt == 6)
    r = rt.Runtime(fmt, prog)
    whole = r.run_block(x, C, C)
    st_whole = r.sync_state().copy()
    r.release()
    r = rt.Runtime(fmt, prog)
    parts = np.concatenate([r.run_block(x[a:b], C, C) for a, b in [(0, 256), (256, 512), (512, 513), (513, 1000), (1000, 1024)]])
    assert (whole.view(np.uint32) == parts.view(np.uint32)).all()
    assert (r.sync_state() == st_whole).all()
    pick = [0, 1, 63, 64, 97, 1000, 2047, 2048, 4095, 777, 3333, 15, 16, 17, 255, 256]
    for c in pick:
        sub = pb.ProgramWriter(fmt)
        sub.core(); sub.param()
        bank = sub.biquad_bank(pb.synth_sections(c, S, pb.F48000, pb.F48000))
        sub.load_gain_fixed(1, 1.0); sub.biquads(bank, S); sub.sat0db(); sub.store(0)
        o = po.OracleProgram(fmt, sub.end_of_code())
        want = o.run_block(np.ascontiguousarray(x[:, c:c + 1]), 1, 1)
        assert_close(np.ascontiguousarray(whole[:, c:c + 1]), want, fmt, what=f"channel {c}")


@pytest.mark.parametrize("fir_impl", [1, 0])
def test_full_size_fir_impulse_and_reblocking(fir_impl):
    """cfg4: 256 ch x 4096 taps x 1024 frames, format 6.  (a) an impulse of 2^-2 returns the taps
    scaled by exactly 2^-2 (one non-zero product per output: exact in any summation order);
    (b) re-blocking invariance; (c) linearity in exact arithmetic: doubling the input doubles the
    output bit for bit (power-of-two scaling commutes with every rounding)."""
    C, T, B = 256, 4096, 1024
    taps = pb.lcg_taps_all(C, T)
    prog = pb.synth_program(6, C, 0, T, taps=taps)
    r = rt.Runtime(6, prog)
    r.set_option("fir_impl", fir_impl)
    imp = np.zeros((5 * B, C), dtype=np.float32)
    imp[3, :] = 0.25
    y = np.concatenate([r.run_block(imp[k * B:(k + 1) * B], C, C) for k in range(5)])
    assert (y[:3] == 0).all()
    assert (y[3:3 + T].T.view(np.uint32) == (taps * np.float32(0.25)).view(np.uint32)).all()
    assert (y[3 + T:] == 0).all()
    r.release()
    x = pb.lcg_input(2 * B, C, True, seed=4)
    r = rt.Runtime(6, prog); r.set_option("fir_impl", fir_impl)
    a = np.concatenate([r.run_block(x[:B], C, C), r.run_block(x[B:], C, C)])
    r.release()
    r = rt.Runtime(6, prog); r.set_option("fir_impl", fir_impl)
    b = np.concatenate([r.run_block(x[s:e], C, C) for s, e in [(0, 100), (100, 1124), (1124, 1125), (1125, 2048)]])
    assert_close(b, a, 6, exact=True, what="re-blocked")
    r.release()
    r = rt.Runtime(6, prog); r.set_option("fir_impl", fir_impl)
    c2 = np.concatenate([r.run_block(2 * x[:B], C, C), r.run_block(2 * x[B:], C, C)])
    assert (c2.view(np.uint32) == (2 * a).view(np.uint32)).all()


def test_north_star_program_sampled_oracle():
    """4096 ch x (16 biquads + 4096-tap FIR), format 6, 1024 frames x 2 blocks: sampled channels vs oracle."""
    C, S, T, B = 4096, 16, 4096, 1024
    pick = [0, 1, 2047, 4095]
    taps = pb.lcg_taps_all(C, T)
    prog = pb.synth_program(6, C, S, T, taps=taps)
    x = pb.lcg_input(2 * B, C, True)
    r = rt.Runtime(6, prog)
    got = np.concatenate([r.run_block(x[:B], C, C), r.run_block(x[B:], C, C)])
    for c in pick:
        sub = pb.ProgramWriter(6, capacity=1 << 15)
        sub.core(); sub.param()
        bank = sub.biquad_bank(pb.synth_sections(c, S, pb.F48000, pb.F48000))
        imp = sub.fir_impulses([taps[c]])
        sub.load_gain_fixed(1, 1.0); sub.biquads(bank, S); sub.fir(imp, T); sub.sat0db(); sub.store(0)
        o = po.OracleProgram(6, sub.end_of_code())
        want = o.run_block(np.ascontiguousarray(x[:, c:c + 1]), 1, 1)
        assert_close(np.ascontiguousarray(got[:, c:c + 1]), want, 6, what=f"channel {c}")


@pytest.mark.parametrize("fmt", [3, 5])
@pytest.mark.parametrize("channels,sections,taps", [(1, 1, 0), (70, 5, 0), (130, 16, 33), (9, 0, 120), (3, 24, 7), (4, 1, 2300), (3, 0, 4100), (6, 2, 257)])
def test_float_accumulator_models_on_chain_lane(fmt, channels, sections, taps):
    """DSP_FORMAT 3 and 5 (float accumulator, truncating dspMulFloatFloat): chain programs run one lane per chain, their FIRs one
    lane per (chain, frame) over blocks (fir_lane: tap chunks of 2048, frame tiles of 256) and tap by tap on single frames; ragged
    blocks, state carried, and the same chains cut into shards, against the oracle bit for bit."""
    prog = pb.synth_program(fmt, channels, sections, taps)
    blocks = [1, 7, 64, 100, 33, 2] if taps < 200 else [300, 1, 700, 1, 257, 2]
    x = pb.lcg_input(sum(blocks), channels, fmt == 5, seed=50 + channels)
    r = _oracle_vs_device(fmt, prog, x, channels, channels, blocks)
    assert r.core_info()["chains"] == channels
    if channels >= 3:
        o = po.OracleProgram(fmt, prog)
        want = o.run_block(x, channels, channels)
        r.release()
        r = rt.Runtime(fmt, prog)
        got = np.zeros_like(want)
        for rank in range(3):
            r.set_shard(rank, 3)
            info = r.shard_info()
            lo, n = info["first_chain"], info["nchains"]
            got[:, lo:lo + n] = r.run_block(np.ascontiguousarray(x[:, lo:lo + n]), n, channels + lo, lo)
        r.set_shard(0, 1)
        assert (got.view(np.uint32) == want.view(np.uint32)).all()
        assert (r.sync_state() == o.state).all()


@pytest.mark.parametrize("fmt", [3, 5])
def test_float_accumulator_models_at_the_headline_geometry(fmt):
    """the north-star chain shape (16 sections + 4096 taps, blocks of 1024 frames) on 100 chains in DSP_FORMAT 3 / 5: chain_rows feeding
    fir_lane through the sequence buffer, two full blocks and a ragged one, state carried -- against the oracle bit for bit"""
    C = 100
    prog = pb.synth_program(fmt, C, 16, 4096)
    blocks = [1024, 1024, 333]
    x = pb.lcg_input(sum(blocks), C, fmt == 5, seed=77)
    _oracle_vs_device(fmt, prog, x, C, C, blocks).release()


@pytest.mark.parametrize("sections,taps", [(3, 40), (0, 300), (2, 0)])
def test_float_accumulator_models_with_nan_inf_and_huge_samples(sections, taps):
    """DSP_FORMAT 5 (float samples): Inf, NaN, values next to FLT_MAX and subnormals among the inputs.  chain_rows and fir_lane take
    the software product's operands apart ahead of time and add with the hardware; a product whose exponent field fills up may read as a
    NaN, which the reference adds the SSE way -- those blocks run again through the interpreter's own functions.  Bit for bit against
    the oracle, outputs and state."""
    C = 6
    prog = pb.synth_program(5, C, sections, taps)
    blocks = [64, 200, 1, 90]
    x = pb.lcg_input(sum(blocks), C, True, seed=123)
    odd = np.array([np.inf, -np.inf, np.nan, 3.0e38, -3.3e38, 1e-40, -1e-42, 2.0e38], dtype=np.float32)
    rng = np.random.default_rng(7)
    for k in range(40):
        x[rng.integers(0, x.shape[0]), rng.integers(0, C)] = odd[k % len(odd)]
    x[70:75, 2] = odd[:5]                                     # a run of them inside one FIR window
    _oracle_vs_device(5, prog, x, C, C, blocks).release()


def _hw_boundary_input(n, C, seed):
    """float samples that walk the limits of the hardware-product path of formats 3 / 5 (fir_lane_hw, chain_rows): channels of ordinary
    noise; noise at 1e-12, 1e-22, 1e-30 and 1e-37 (the last ones under the band: the integer products take over, and sums are flushed
    to signed zeros); bursts between runs of +0 and -0; +a / -a pairs that cancel exactly; single impulses that decay into silence."""
    x = pb.lcg_input(n, C, True, seed=seed)
    rng = np.random.default_rng(seed)
    scale = [1.0, 1e-12, 1e-22, 1e-30, 1e-37, 1.0, 1.0, 1.0]
    for c in range(C):
        x[:, c] *= np.float32(scale[c % len(scale)])
    xv = x.view(np.uint32)
    if C > 5:
        x[:, 5] = 0.0
        x[::97, 5] = np.float32(0.25)                            # impulses, a decay behind each
        xv[3::97, 5] = 0x80000000                                # ... and -0 samples
    if C > 6:
        a = rng.standard_normal(n).astype(np.float32)
        x[0::2, 6] = a[0::2][: len(x[0::2, 6])]
        x[1::2, 6] = -x[0::2, 6][: len(x[1::2, 6])]             # pairs that cancel
    if C > 7:
        x[:, 7] = 0.0
        x[n // 3: n // 3 + 40, 7] = rng.standard_normal(40).astype(np.float32) * np.float32(1e-33)
    return x


@pytest.mark.parametrize("lane_hw", [1, 0])
@pytest.mark.parametrize("sections,taps", [(0, 300), (0, 2500), (4, 0), (16, 0), (3, 700)])
def test_float_accumulator_models_on_the_limits_of_the_hardware_product(sections, taps, lane_hw):
    """DSP_FORMAT 5 with the products of the tap loop and of the section update made by v_mul_f32 under round-toward-zero where the
    operands' exponents make that dspMulFloatFloat bit for bit (dsp_ieee754.h:335-375; tools/rtz_mul_probe.hip), by the integer
    restatement elsewhere -- inputs on both sides of every condition of that choice, taps over thirty orders of magnitude; the same
    with the option off.  Against the oracle bit for bit, outputs and state."""
    C = 8
    blocks = [700, 1, 64, 1024, 300]
    n = sum(blocks)
    tp = None
    if taps:
        rng = np.random.default_rng(99)
        tp = pb.lcg_taps_all(C, taps).astype(np.float32)
        tp[1] *= np.float32(1e-12)
        tp[2] *= (10.0 ** rng.uniform(-30, 0, taps)).astype(np.float32)
        tp[3] *= (10.0 ** rng.uniform(-38, -20, taps)).astype(np.float32)
        tp[4, ::3] = 0.0
        tp[6] = np.float32(1.0)                                   # equal taps: the cancelling pairs sum to exact zeros
    prog = pb.synth_program(5, C, sections, taps, taps=tp)
    x = _hw_boundary_input(n, C, seed=31 + sections + taps)
    _oracle_vs_device(5, prog, x, C, C, blocks, opts={"lane_hw": lane_hw}).release()
    rt.Runtime.set_global_option("lane_hw", 1)


@pytest.mark.parametrize("gain", [1.0, 1e-9, 3e-31])
def test_int_sample_float_accumulator_model_with_small_gains(gain):
    """DSP_FORMAT 3 (int samples, float accumulator): LOAD_GAIN with gains that put the cascade's and the FIR's operands in the
    band, at its edge and under it"""
    C = 5
    prog = pb.synth_program(3, C, 6, 500, gain=gain)
    blocks = [600, 1, 1024, 77]
    x = pb.lcg_input(sum(blocks), C, False, seed=5)
    x[100:400, 2] = 0
    _oracle_vs_device(3, prog, x, C, C, blocks).release()


@pytest.mark.parametrize("fmt", [4, 6])
@pytest.mark.parametrize("sections", [1, 3, 16])
def test_signed_zeros_in_the_state_across_one_frame_blocks(fmt, sections):
    """-0 and negative subnormal samples, blocks of one frame between longer ones: a one-frame block hands x1 / y1 on as x2 / y2
    unchanged, sign of zero included (tests/dev/gpu_fuzz_sweep.py seed 9008 found biquad_row writing +0 there).  The state is
    compared with the oracle's after every block."""
    C = 5
    prog = pb.synth_program(fmt, C, sections, 0)
    blocks = [64, 1, 1, 7, 1, 33, 1]
    x = pb.lcg_input(sum(blocks), C, fmt == 6, seed=9)
    if fmt == 6:
        xv = x.view(np.uint32)
        xv[60:72:2, :] = 0x80000000                           # -0
        xv[61:71:4, 1] = 0x802D27B4                           # a negative subnormal
        xv[100:108, 3] = 0x80000000
    else:
        x[60:72:2, :] = 0
    o = po.OracleProgram(fmt, prog)
    r = rt.Runtime(fmt, prog)
    pos = 0
    for b in blocks:
        want = o.run_block(x[pos:pos + b], C, C)
        got = r.run_block(x[pos:pos + b], C, C)
        assert (got.view(np.uint32) == want.view(np.uint32)).all(), f"block at frame {pos}"
        assert (r.sync_state() == o.state).all(), f"state after the block at frame {pos} ({b} frames)"
        pos += b
    r.release()


@pytest.mark.parametrize("fmt", [2, 4, 6])
@pytest.mark.parametrize("fanout", [1, 0])
def test_cascades_of_many_lengths_in_one_launch(fmt, fanout):
    """A core whose chains have DIFFERENT section counts -- a real crossover -- used to be one cascade launch per count, one after the
    other (1024 chains with 1 .. 8 sections: 248 us per block).  Round 5: all cascades of up to 16 sections are ONE biquad_row launch
    (a table of rows of every length, four-row waves of one length each), longer ones go out side by side over streams.  Here: 37 chains
    with 1 .. 16, 17, 20 and 40 sections (the last three: biquad_pipe groups beside the merged launch), ragged blocks, against the oracle --
    and the same with "group_fanout" 0 (one launch per length, as before)."""
    from avdsp_amd import encoder as enc
    import ctypes as C
    counts = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 20, 40, 3, 3, 8, 8, 8, 16, 16, 1, 2, 12, 12, 12, 12, 12, 5, 6, 7, 9]
    n = len(counts)

    def build(L):
        banks = []
        for c in range(n):
            L.dsp_PARAM()
            banks.append(L.dspBiquad_Sections(counts[c]))
            for b in range(counts[c]):
                L.dsp_Filter2ndOrder(9, C.c_double(100.0 + 37 * b + 3 * c), C.c_double(0.7 + 0.05 * (b % 5)), C.c_float(1.2 if b & 1 else 0.8))
        L.dsp_CORE()
        for c in range(n):
            L.dsp_LOAD_GAIN_Fixed(n + c, C.c_float(0.5)); L.dsp_BIQUADS(banks[c]); L.dsp_SAT0DB(); L.dsp_STORE(c)
    prog = enc.encode(build, 2 if fmt == 2 else 6, pb.F48000, pb.F48000, max_io=2 * n + 8, capacity=1 << 20)
    x = pb.lcg_input(2300, n, fmt == 6, seed=fmt)
    o = po.OracleProgram(fmt, prog)
    r = rt.Runtime(fmt, prog)
    r.set_option("group_fanout", fanout)
    assert r.core_info(0)["chains"] == n
    try:
        pos = 0
        for b in (1024, 37, 700, 1, 538):
            want, got = o.run_block(x[pos:pos + b], n, n), r.run_block(x[pos:pos + b], n, n)
            bad = np.nonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=0))[0]
            assert bad.size == 0, f"block at {pos}: chains {bad.tolist()} (sections {[counts[i] for i in bad]}) differ"
            pos += b
        assert (r.sync_state() == o.state).all()
    finally:
        r.set_option("group_fanout", 1)
        r.release()


@pytest.mark.parametrize("fmt", [2, 4, 6])
@pytest.mark.parametrize("biquad_impl", [1, 0])
def test_cascades_longer_than_a_wave_run_as_pieces(fmt, biquad_impl):
    """More than 64 sections in one chain do not fit a wave's lanes: until round 5 such a chain fell to biquad_simple (a lane per chain,
    state in memory -- 4096 chains x 65 sections 81 ms against 136 us for 64).  Now add_plan cuts it into pieces of up to 64 sections that
    run as launches one after the other, the 32-bit word between two sections going through a scratch column.  Chains of 65, 100, 128, 129
    and 200 sections beside short ones, plain and gain loads, with and without SAT0DB, two of them in front of a FIR, ragged blocks (one
    of more than 1024 frames) against the oracle, outputs and state; in format 6 then an Inf and a NaN sample (the pieces' replay in the
    reference's own order).  biquad_impl 0: the same pieces through biquad_simple."""
    counts = [65, 100, 128, 129, 200, 16, 3, 64, 65, 65]
    n = len(counts)
    w = pb.ProgramWriter(fmt, pb.F48000, pb.F48000, capacity=1 << 16)
    w.core()
    for c, S in enumerate(counts):
        w.param()
        bank = w.biquad_bank(pb.synth_sections(c, S, pb.F48000, pb.F48000))
        T = 40 if (fmt != 2 and c in (1, 3)) else 0
        imp = w.fir_impulses([pb.lcg_taps(c, T)]) if T else None
        if c & 1:
            w.load_gain_fixed(n + c, 0.5)
        else:
            w.load(n + c)
        w.biquads(bank, S)
        if T:
            w.fir(imp, T)
        if c % 3:
            w.sat0db()
        w.store(c)
    prog = w.end_of_code()
    x = pb.lcg_input(3200, n, fmt == 6, seed=10 + fmt)
    if fmt == 6:
        x[2900, 0] = np.inf; x[2950, 4] = np.nan; x[3000, 8] = -np.inf
    o = po.OracleProgram(fmt, prog)
    r = rt.Runtime(fmt, prog)
    r.set_option("biquad_impl", biquad_impl)
    assert r.core_info(0)["chains"] == n
    try:
        pos = 0
        for b in (1024, 37, 1500, 1, 638):
            want, got = o.run_block(x[pos:pos + b], n, n), r.run_block(x[pos:pos + b], n, n)
            bad = np.nonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=0))[0]
            assert bad.size == 0, f"block at {pos}: chains {bad.tolist()} (sections {[counts[i] for i in bad]}) differ"
            pos += b
        assert (r.sync_state() == o.state).all()
    finally:
        r.set_option("biquad_impl", 1)
        r.release()
