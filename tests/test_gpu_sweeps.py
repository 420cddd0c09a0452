"""A slice of the development sweeps (tests/dev/) inside the -m gpu suite: random chain shapes through the parallel kernels in all five
formats, and cores ending in runs of identical strands of random shapes through the strand plans and the interpreter's strand groups --
each against the oracle, outputs and final state, bit for bit.  The full sweeps (thousands of seeds) stay a builder's tool; what they
find is promoted into tests of its own (tests/test_gpu_parity.py::test_signed_zeros_in_the_state_across_one_frame_blocks)."""
import pytest

from avdsp_amd import runtime as rt

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _release():
    yield
    rt.lib().dspRuntimeSetOption(b"strand_lanes", 1)
    rt.lib().dspRuntimeRelease()


@pytest.mark.parametrize("lo", range(0, 200, 20))
def test_random_chain_shapes_slice(lo):
    """seeds lo .. lo + 19 of tests/dev/gpu_chain_sweep.py, formats 2 .. 6 (up to 40 channels, 70 sections, 1500 taps, 2500 frames)"""
    from tests.dev.gpu_chain_sweep import run
    n, bad = run(lo, lo + 20, formats=(2, 3, 4, 5, 6))
    assert n > 0 and not bad, bad


@pytest.mark.parametrize("lo", range(0, 200, 20))
def test_random_strand_shapes_slice(lo):
    """seeds lo .. lo + 19 of tests/dev/gpu_strand_sweep.py: five formats x strand plan / strand groups"""
    from tests.dev.gpu_strand_sweep import run
    n, bad, _ = run(lo, lo + 20)
    assert n > 0 and not bad, bad


@pytest.mark.parametrize("lo", [0, 240, 260, 280, 300, 480, 560, 620, 820, 880, 940])      # (seeds 241 ... 956: programs with a frame-by-frame piece beside frame-parallel ones,
def test_random_programs_with_windows_that_share_io_numbers(lo):                          #  which a first version of show_through got wrong: a whole-row write-back beside its neighbours)
    """random multi-core programs (tests/fuzz_programs.py) through dspRuntimeBlockAll with an output window IO 0 .. 47 that contains
    the input window IO 32 .. 39: the shared columns show the input unless the program stores them (the reference's one samples[]
    frame; on the device: show_through in front of the call's launches, every launch moves its core's slots only).  Five formats,
    blocks of 2 / 64 / all frames, outputs and the whole data area against the oracle."""
    import numpy as np
    from avdsp_amd import progbuilder as pb
    from oracle import pyoracle as po
    from tests.fuzz_programs import IN_BASE, N_IN, random_program
    frames, out_stride = 200, 48
    n = 0
    for seed in range(lo, lo + 20):
        for fmt in (2, 3, 4, 5, 6):
            prog = random_program(seed, fmt)
            fs, block = [48000, 48000, 96000][seed % 3], [2, 64, frames][seed % 3]
            x = pb.lcg_input(frames, N_IN, fmt in (5, 6), seed=seed)
            r = rt.Runtime(fmt, prog, fs=fs, random=seed, dither=24)
            if r.rc < 0:
                continue
            o = po.OracleProgram(fmt, prog, fs=fs, random=seed, dither=24)
            want = o.run_block(x, out_stride, IN_BASE, 0, block=block, frame=np.zeros(4096, dtype=np.uint32))
            got = r.run_block_all(x, out_stride, IN_BASE, 0, block=block)
            r.sync_state()
            nn = int(prog[1]) + int(prog[2])
            cols = np.nonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=0))[0]
            assert cols.size == 0, f"seed {seed} DSP_FORMAT {fmt} block {block}: output columns {list(cols)} differ"
            assert (r.buf[12:nn] == o.buf[12:nn]).all(), f"seed {seed} DSP_FORMAT {fmt}: data area differs"
            r.release()
            n += 1
    assert n > 0


@pytest.mark.parametrize("lo", [0, 18, 36])
def test_overlap_arrangements_on_random_shapes(lo):
    """seeds lo .. lo + 17 of tests/dev/gpu_overlap_sweep.py: random cascade + FIR programs, 4 .. 9 device-resident blocks enqueued back to
    back, through the overlap mode's arrangements in turn ("ready_words" 0 / 1 / 2 / by plan, "ring_wait" 0 / 1, "overlap" 2, both FIR
    boundaries, fir_flow); every block's outputs and the final state against the oracle, and no bounded wait ran out"""
    from tests.dev.gpu_overlap_sweep import run
    n, bad = run(lo, lo + 18)
    assert n == 18 and not bad, bad


@pytest.mark.parametrize("lo", range(0, 60, 20))
def test_random_chain_programs_in_instances_slice(lo):
    """seeds lo .. lo + 19 of tests/dev/gpu_instance_sweep.py: random chain programs (cascades, FIRs, both; five formats in turn) in 1 .. 37
    instances, ragged blocks, every instance with its own input against the oracle -- outputs and each instance's state"""
    from tests.dev.gpu_instance_sweep import run
    n, bad = run(lo, lo + 20)
    assert n == 20 and not bad, bad


@pytest.mark.parametrize("rows", [2, 4])
def test_random_chain_shapes_with_forced_row_tiles(rows):
    """seeds 200 .. 239 of the chain sweep with fir_tile's row tiles forced: short and ragged blocks (1 .. 2500 frames) then regroup a
    workgroup's waves by the tiles the block has (round 5), including four chains of one 512-frame tile each"""
    import os
    from tests.dev.gpu_chain_sweep import run
    os.environ["AVDSP_SWEEP_OPTIONS"] = f"fir_rows={rows}"
    try:
        n, bad = run(200, 240, formats=(4, 6))
    finally:
        os.environ.pop("AVDSP_SWEEP_OPTIONS", None)
    assert n > 0 and not bad, bad


def test_many_chains_in_short_and_ragged_blocks_slice():
    """seeds 0 .. 9 of tests/dev/gpu_wide_blocks_sweep.py: 300 .. 3000 chains (two and four row tiles per FIR wave by the automatic choice)
    in blocks of 1 .. 1024 frames -- the regrouping of a workgroup's waves by the tiles a block has, where the chain count makes it matter"""
    from tests.dev.gpu_wide_blocks_sweep import run
    n, bad = run(0, 10)
    assert n == 10 and not bad, bad
