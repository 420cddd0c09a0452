"""fir_tile<FMT, R> (one wave per channel tile of 256 R frames, R row tiles sharing each taps operand; no workgroup barrier in
the tap loop) against the oracle, bit for bit, for every R, and against round 1's fir_mfma and the plain tap loop.  Tap counts
around the kernel's seams (k-steps of 4, unrolled groups of 16 k-steps, chunks of 160 / 208 / 256 k-steps), ragged blocks,
FIR-only chains (the kernel appends their input itself) and chains behind a cascade, both float models."""
import numpy as np
import pytest

from avdsp_amd import progbuilder as pb
from avdsp_amd import runtime as rt
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _release():
    yield
    rt.lib().dspRuntimeSetOption(b"fir_rows", 0)
    rt.lib().dspRuntimeSetOption(b"fir_split", 0)
    rt.lib().dspRuntimeSetOption(b"fir_lean", -1)
    rt.lib().dspRuntimeRelease()


def words(a):
    return np.ascontiguousarray(a).view(np.uint32)


def run_vs_oracle(fmt, prog, x, C, blocks, rows, fir_impl=1, fir_lean=None):
    o = po.OracleProgram(fmt, prog)
    r = rt.Runtime(fmt, prog)
    r.set_option("fir_impl", fir_impl)
    r.set_option("fir_rows", rows)
    if fir_lean is not None:
        r.set_option("fir_lean", fir_lean)
    pos = 0
    for b in blocks:
        want = o.run_block(x[pos:pos + b], C, C)
        got = r.run_block(x[pos:pos + b], C, C)
        bad = np.nonzero((words(got) != words(want)).any(axis=0))[0]
        assert bad.size == 0, f"rows {rows}: block at frame {pos} ({b} frames): channels {bad[:8].tolist()} differ, first frame " \
                              f"{int(np.argmax((words(got) != words(want)).any(axis=1)))}"
        pos += b
    assert (r.sync_state() == o.state).all(), "state differs after the last block"
    r.set_option("fir_impl", 1)


ROWS = [1, 2, 4]                 # row tiles per wave
IMPLS = [4, 3, 1]                # fir_flow, fir_stream, fir_tile


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("rows", ROWS)
@pytest.mark.parametrize("taps", [1, 3, 4, 7, 60, 61, 64, 65, 255, 256, 257, 580, 641, 1000, 1030])
def test_fir_only_chains_every_row_count(rows, taps, impl):
    C = 5
    prog = pb.synth_program(6, C, 0, taps)
    blocks = [1024, 1, 37, 256, 257, 700, 1024, 513]
    x = pb.lcg_input(sum(blocks), C, True, seed=taps)
    run_vs_oracle(6, prog, x, C, blocks, rows, impl)


@pytest.mark.parametrize("impl", IMPLS)
@pytest.mark.parametrize("rows", ROWS)
@pytest.mark.parametrize("fmt,C,S,T", [(6, 9, 3, 2048), (6, 3, 16, 4096), (4, 6, 2, 4100), (6, 2, 1, 5000), (4, 4, 0, 2560), (6, 21, 5, 130)])
def test_long_fir_behind_a_cascade(rows, fmt, C, S, T, impl):
    prog = pb.synth_program(fmt, C, S, T)
    blocks = [1024, 1024, 300, 1024, 1024, 1024, 724]
    x = pb.lcg_input(sum(blocks), C, fmt == 6, seed=C + T)
    run_vs_oracle(fmt, prog, x, C, blocks, rows, impl)


@pytest.mark.parametrize("impl", [4, 1])
@pytest.mark.parametrize("fmt,C,S,T", [(6, 5, 0, 1), (6, 3, 0, 255), (6, 4, 0, 893), (6, 6, 0, 3581), (6, 5, 0, 4096), (6, 3, 2, 5000), (4, 4, 0, 2560), (4, 5, 3, 4100)])
def test_long_chunks_where_a_launch_leaves_a_simd_one_wave(fmt, C, S, T, impl):
    """automatic row choice with few chains: fir_tile's chunks of 320 k-steps, fir_flow's of 224 with two window images (the tap
    counts sit on both sides of one, two and several such chunks); ragged blocks, a FIR-only program and one behind a cascade"""
    prog = pb.synth_program(fmt, C, S, T)
    blocks = [1024, 1024, 1, 300, 1024, 257, 1024, 700]
    x = pb.lcg_input(sum(blocks), C, fmt == 6, seed=T + C)
    run_vs_oracle(fmt, prog, x, C, blocks, 0, impl)


@pytest.mark.parametrize("rows", [0] + ROWS)
def test_many_channels_auto_rows_and_other_kernels_agree(rows):
    """enough chains for every automatic choice (>= 2048: 4 row tiles), short taps so that the oracle keeps up; the same blocks
    through fir_mfma (fir_impl 2) and the plain tap loop (fir_impl 0) must give the same bits"""
    C, T = 2100, 70
    taps = pb.lcg_taps_all(C, T)
    prog = pb.synth_program(6, C, 0, T, taps=taps)
    x = pb.lcg_input(1024 + 500, C, True, seed=3)
    o = po.OracleProgram(6, prog)
    want = np.concatenate([o.run_block(x[:1024], C, C), o.run_block(x[1024:], C, C)])
    for impl in ((4, 3, 1, 2, 0) if rows == 0 else (4, 3, 1)):
        r = rt.Runtime(6, prog)
        r.set_option("fir_impl", impl)
        r.set_option("fir_rows", rows)
        got = np.concatenate([r.run_block(x[:1024], C, C), r.run_block(x[1024:], C, C)])
        assert (words(got) == words(want)).all(), f"fir_impl {impl}"
        assert (r.sync_state() == o.state).all()
        r.set_option("fir_impl", 1)
        r.release()


def test_nan_inf_and_subnormal_samples_in_the_window():
    """the window is converted while it is staged: exponent 255 reads as 1.m x 2^128 and subnormals as zero, like the reference's
    bit-field product (dsp_ieee754.h:377-410)"""
    C, T = 3, 300
    prog = pb.synth_program(6, C, 0, T)
    x = pb.lcg_input(2048, C, True, seed=11)
    xi = x.view(np.uint32)
    xi[5, 0] = 0x7F800000; xi[9, 1] = 0xFFC00001; xi[700, 2] = 0x7F812345; xi[1030, 0] = 0x00000012; xi[1500, 1] = 0x80000400
    for rows in ROWS:
        for impl in IMPLS:
            run_vs_oracle(6, prog, x, C, [1024, 1024], rows, impl)


def _program_with_out_map(C, T, out_of):
    """FIR-only chains like synth_program's, but chain c stores to IO out_of(c)"""
    pw = pb.ProgramWriter(6, pb.F48000, pb.F48000, capacity=64 + C * (T + 64))
    taps = pb.lcg_taps_all(C, T)
    pw.core()
    for c in range(C):
        pw.param()
        imp = pw.fir_impulses([taps[c]])
        pw.load_gain_fixed(C + c, 1.0)
        pw.fir(imp, T)
        pw.sat0db()
        pw.store(out_of(c))
    return pw.end_of_code()


@pytest.mark.parametrize("layout", ["reversed", "shifted_window", "aligned"])
def test_four_row_tiles_store_together_only_when_the_columns_allow(layout):
    """fir_tile at four row tiles sends the workgroup's four channels as 16-byte stores when their output columns are consecutive
    and aligned; chains stored in reverse order, or an output window that starts one IO off a multiple of four, take the plain way"""
    C, T, B = 2052, 40, 1024
    out_of = (lambda c: C - 1 - c) if layout == "reversed" else (lambda c: c)
    prog = _program_with_out_map(C, T, out_of)
    x = pb.lcg_input(B, C, True, seed=17)
    o = po.OracleProgram(6, prog)
    r = rt.Runtime(6, prog)
    r.set_option("fir_rows", 4)
    if layout == "shifted_window":
        # the caller's output window starts at IO -1 ... i.e. one spare column in front: column of chain 0 is misaligned
        want = o.run_block(x, C, C)
        out = np.zeros((B, C + 1), dtype=x.dtype)
        f = getattr(r.L, "dspRuntimeBlock_6")
        # window [frames][C + 1] laid over IOs 0 .. C: base pointer one word in, stride C + 1
        view = out.reshape(-1)[1:]
        rc = f(r.cores[0], r.rundata, x.ctypes.data, C, C, view.ctypes.data, C + 1, 0, B)
        assert rc >= 0, r.last_error()
        got = np.lib.stride_tricks.as_strided(view, shape=(B, C), strides=((C + 1) * 4, 4))
        assert (words(np.ascontiguousarray(got)) == words(want)).all()
    else:
        want = o.run_block(x, C, C)
        got = r.run_block(x, C, C)
        assert (words(got) == words(want)).all()
    assert (r.sync_state() == o.state).all()


@pytest.mark.parametrize("fmt,C,S,T", [(6, 5, 0, 4096), (6, 3, 0, 1), (6, 4, 0, 37), (6, 6, 2, 4100), (4, 4, 0, 2560), (6, 5, 0, 641)])
def test_opt_in_tap_split_is_within_the_stated_tolerance(fmt, C, S, T):
    """dspRuntimeSetOption("fir_split", 1), off by default: a fir_tile launch that leaves a SIMD one wave at most (here: always)
    cuts every tile's taps over two waves and adds the two partial sums -- (taps 0 .. S/2) + (taps S/2 ..), NOT the reference's
    summation order, so the check is BASELINE's float-mode tolerance (1e-6 of the block's peak), not bit equality; the delay lines
    (the state) are the reference's bits all the same, and the option off gives the reference's bits again."""
    prog = pb.synth_program(fmt, C, S, T)
    blocks = [1024, 300, 1, 1024, 513]
    x = pb.lcg_input(sum(blocks), C, fmt == 6, seed=C + T)
    o = po.OracleProgram(fmt, prog)
    want = np.concatenate([o.run_block(x[p:p + b], C, C) for p, b in zip(np.cumsum([0] + blocks[:-1]), blocks)])
    r = rt.Runtime(fmt, prog)
    r.set_option("fir_split", 1)
    assert r.get_option("fir_split") == 1
    got = np.concatenate([r.run_block(x[p:p + b], C, C) for p, b in zip(np.cumsum([0] + blocks[:-1]), blocks)])
    if fmt == 6:
        err = np.abs(got.astype(np.float64) - want.astype(np.float64)).max()
        peak = np.abs(want.astype(np.float64)).max()
    else:
        err = np.abs(got.astype(np.int64) - want.astype(np.int64)).max()
        peak = np.abs(want.astype(np.int64)).max()
    assert err <= 1e-6 * peak, f"max abs error {err} against a peak of {peak}"
    assert (r.sync_state() == o.state).all(), "the delay lines do not depend on the summation order"
    r.set_option("fir_split", 0)
    r.release()
    r = rt.Runtime(fmt, prog)
    got = np.concatenate([r.run_block(x[p:p + b], C, C) for p, b in zip(np.cumsum([0] + blocks[:-1]), blocks)])
    assert (words(got) == words(want)).all()


@pytest.mark.parametrize("fir_lean", [0, 1])
@pytest.mark.parametrize("rows", ROWS)
@pytest.mark.parametrize("fmt,C,S,T", [(6, 5, 0, 641), (6, 7, 3, 2048), (4, 6, 2, 4100), (6, 3, 0, 4096)])
def test_both_chunk_boundaries_every_row_count(rows, fmt, C, S, T, fir_lean):
    """fir_tile's long chunk boundary (a masked offset and a class look per window sample) and the lean one (one masked offset per
    lane and chunk into a ring that holds every sample twice; Inf / NaN found in the tile's sums at the end), forced either way
    ("fir_lean" 0 / 1; the library chooses by plan): ragged blocks -- the ring wraps several times --, with Inf, NaN, subnormal and
    exponent-255 samples in some channels (float samples), against the oracle bit for bit, outputs and state."""
    prog = pb.synth_program(fmt, C, S, T)
    blocks = [1024, 1, 37, 1024, 300, 1024, 1024, 513, 1024, 1024, 1024]
    x = pb.lcg_input(sum(blocks), C, fmt == 6, seed=T + rows)
    if fmt == 6:
        xv = x.view(np.uint32)
        xv[5, 0] = 0x7F800000; xv[1030, 1] = 0xFFC00001; xv[2100, 2] = 0x7F7FFFFF; xv[2101, 2] = 0xFF7FFFFF
        xv[3000:3004, 1] = 0x00012345; xv[3500, 0] = 0x80000000; xv[4000, C - 1] = 0x7F812345
    run_vs_oracle(fmt, prog, x, C, blocks, rows, 1, fir_lean)


@pytest.mark.parametrize("fir_lean", [0, 1])
@pytest.mark.parametrize("rows", ROWS)
@pytest.mark.parametrize("fmt,C,S,T", [(6, 8, 0, 300), (6, 9, 2, 700), (4, 12, 3, 130), (6, 5, 0, 2048)])
def test_short_blocks_take_as_many_tiles_as_they_have(rows, fmt, C, S, T, fir_lean):
    """Round 5: a chain's waves in a launch are the tiles its block HAS -- 1, 2 or 4 of a one-row-tile wave's four, 1 or 2 of a
    two-row-tile wave's two -- not the four quarters of a 1024-frame block (three of which left at once on a 256-frame block while
    their workgroup kept its LDS: two live waves per CU).  Block lengths on both sides of every tile count, including the case that
    regroups a two-row-tile workgroup as FOUR chains with one 512-frame tile each (their tiles leave together as 16-byte pieces when
    the columns allow: 8 and 12 chains do, 9 and 5 leave some chains to the plain way)."""
    prog = pb.synth_program(fmt, C, S, T)
    blocks = [64, 256, 257, 512, 511, 384, 128, 513, 1024, 768, 769, 300, 1, 255]
    x = pb.lcg_input(sum(blocks), C, fmt == 6, seed=T + C + rows)
    run_vs_oracle(fmt, prog, x, C, blocks, rows, 1, fir_lean)
