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
