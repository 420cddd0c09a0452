"""dspRuntimeBlockSubmit / dspRuntimeBlockWait: host-pointer blocks as a queue (copies of one block under the kernels of another)
give the bits of the same dspRuntimeBlock_N calls in the same order -- against the oracle, block by block, state included."""
import numpy as np
import pytest

from avdsp_amd import progbuilder as pb
from avdsp_amd import runtime as rt
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _release():
    yield
    rt.lib().dspRuntimeRelease()


def words(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("fmt,C,S,T,B", [(6, 64, 2, 300, 1024), (6, 33, 0, 1000, 512), (4, 16, 3, 70, 1024), (2, 40, 4, 0, 256)])
@pytest.mark.parametrize("depth", [1, 2, 4, 6])
@pytest.mark.parametrize("overlap", [0, 1])
def test_queued_blocks_match_the_oracle(fmt, C, S, T, B, depth, overlap):
    prog = pb.synth_program(fmt, C, S, T)
    nblk = 9
    x = pb.lcg_input(B * nblk, C, fmt in (5, 6), seed=C + T)
    o = po.OracleProgram(fmt, prog)
    r = rt.Runtime(fmt, prog)
    r.set_option("overlap", overlap)                          # the next block's cascade under this block's FIR: it must wait for its copy
    ring = 8                                                  # buffers the host cycles through (>= depth + what it still reads)
    xin = [np.empty((B, C), dtype=x.dtype) for _ in range(ring)]
    out = [np.zeros((B, C), dtype=x.dtype) for _ in range(ring)]
    done = 0
    for k in range(nblk):
        xin[k % ring][:] = x[k * B:(k + 1) * B]
        in_flight = r.submit_block(xin[k % ring], out[k % ring], C, 0)
        assert 1 <= in_flight <= 4
        left = r.wait_blocks(depth - 1)
        assert left <= depth - 1
        while done < k + 1 - left:                            # blocks complete in order
            want = o.run_block(x[done * B:(done + 1) * B], C, C)
            assert (words(out[done % ring]) == words(want)).all(), f"block {done}"
            done += 1
    assert r.wait_blocks(0) == 0
    while done < nblk:
        want = o.run_block(x[done * B:(done + 1) * B], C, C)
        assert (words(out[done % ring]) == words(want)).all(), f"block {done}"
        done += 1
    assert (r.sync_state() == o.state).all()
    r.set_option("host_pin", 0)                               # the buffers are about to be freed
    r.set_option("overlap", 0)


def test_sync_calls_and_small_blocks_between_queued_ones():
    """a block under 256 frames is done on the spot, a synchronous dspRuntimeBlock_N waits for the queue: order is kept"""
    fmt, C, S, T = 6, 12, 2, 130
    prog = pb.synth_program(fmt, C, S, T)
    sizes = [1024, 100, 512, 1, 1024, 300]
    x = pb.lcg_input(sum(sizes), C, True, seed=5)
    o = po.OracleProgram(fmt, prog)
    r = rt.Runtime(fmt, prog)
    bufs, pos = [], 0
    for i, n in enumerate(sizes):
        xi = np.ascontiguousarray(x[pos:pos + n]); yi = np.zeros((n, C), dtype=x.dtype)
        bufs.append((xi, yi, pos, n))
        if i == 4:
            r.run_block(xi, C, C, out=yi)                     # synchronous in the middle
        else:
            r.submit_block(xi, yi, C, 0)
        pos += n
    assert r.wait_blocks(0) == 0
    for xi, yi, p, n in bufs:
        assert (words(yi) == words(o.run_block(x[p:p + n], C, C))).all(), f"block at {p}"
    assert (r.sync_state() == o.state).all()
    r.set_option("host_pin", 0)


def test_interpreter_core_through_submit():
    """with the chain lowering switched off the core runs through the interpreter; submit then does the block on the spot"""
    fmt, C, S, T = 4, 6, 2, 40
    prog = pb.synth_program(fmt, C, S, T)
    x = pb.lcg_input(3 * 512, C, False, seed=9)
    o = po.OracleProgram(fmt, prog)
    r = rt.Runtime(fmt, prog)
    r.set_option("generic", 1)
    outs = []
    for k in range(3):
        xi = np.ascontiguousarray(x[k * 512:(k + 1) * 512]); yi = np.zeros((512, C), dtype=x.dtype)
        r.submit_block(xi, yi, C, 0)
        outs.append(yi)
    assert r.wait_blocks(0) == 0
    for k in range(3):
        assert (words(outs[k]) == words(o.run_block(x[k * 512:(k + 1) * 512], C, C))).all()
    assert (r.sync_state() == o.state).all()
    r.set_option("generic", 0)
