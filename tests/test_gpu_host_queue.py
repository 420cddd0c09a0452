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


def _pack(x: np.ndarray, pcm: int) -> np.ndarray:
    if pcm == rt.PCM_S16:
        return (x >> 16).astype("<i2").view(np.uint8).reshape(-1)
    u = x.view(np.uint32).reshape(-1)
    b = np.empty((u.size, 3), dtype=np.uint8)
    b[:, 0] = (u >> 8) & 0xFF; b[:, 1] = (u >> 16) & 0xFF; b[:, 2] = (u >> 24) & 0xFF
    return b.reshape(-1)


def _unpack_like_the_plugin(raw: np.ndarray, pcm: int) -> np.ndarray:
    if pcm == rt.PCM_S16:
        return raw.view("<i2").astype(np.int32) << 16
    b = raw.reshape(-1, 3).astype(np.uint32)
    return ((b[:, 0] << 8) | (b[:, 1] << 16) | (b[:, 2] << 24)).view(np.int32)


@pytest.mark.parametrize("path", ["block", "pin_split", "pcm16", "pcm24"])
def test_overlap_mode_waits_for_the_librarys_own_producers(path):
    """"overlap" runs the cascade on a stream of its own: behind the synchronous host call's piecewise copies (with and without
    host_pin + host_split) and behind pcm_unpack it must still see the block it is given -- bit identity with the oracle over
    several blocks, the input buffer rewritten between them"""
    pcm = {"pcm16": rt.PCM_S16, "pcm24": rt.PCM_S24_3LE}.get(path)
    fmt = 4 if pcm is not None else 6
    C, S, T, B, nblk = 192, 3, 200, 1024, 5
    prog = pb.synth_program(fmt, C, S, T)
    x = pb.lcg_input(B * nblk, C, fmt == 6, seed=77)
    o = po.OracleProgram(fmt, prog)
    r = rt.Runtime(fmt, prog)
    r.set_option("overlap", 1)
    if path == "pin_split":
        r.set_option("host_pin", 1); r.set_option("host_split", 256)
    xin = np.empty((B, C), dtype=x.dtype)                      # one buffer, refilled per block like a host's period buffer
    yout = np.zeros((B, C), dtype=x.dtype)
    for k in range(nblk):
        xb = x[k * B:(k + 1) * B]
        if pcm is None:
            xin[:] = xb
            r.run_block(xin, C, C, out=yout)
            got, want = yout, o.run_block(xb, C, C)
        else:
            raw = _pack(xb, pcm)
            got = r.run_block_pcm(pcm, raw, B, C, C, C)
            want = o.run_block(_unpack_like_the_plugin(raw, pcm).reshape(B, C), C, C)
        assert (words(got) == words(want)).all(), f"{path}: block {k}"
    assert (r.sync_state() == o.state).all()
    r.set_option("host_pin", 0); r.set_option("host_split", 0); r.set_option("overlap", 0)


def test_fresh_buffers_per_block_end_their_registration_with_the_block():
    """without host_pin a queued block's buffers are registered for the time the block is in flight only: a host that
    allocates new arrays per block and drops them once the block is back (addresses get reused) stays correct"""
    fmt, C, S, T, B = 6, 48, 2, 90, 512
    prog = pb.synth_program(fmt, C, S, T)
    nblk = 24
    x = pb.lcg_input(B * nblk, C, True, seed=31)
    o = po.OracleProgram(fmt, prog)
    r = rt.Runtime(fmt, prog)
    assert r.L.dspRuntimeGetOption(b"host_pin") == 0
    inflight = []
    for k in range(nblk):
        xi = x[k * B:(k + 1) * B].copy(); yi = np.zeros((B, C), dtype=x.dtype)
        r.submit_block(xi, yi, C, 0)
        inflight.append((k, xi, yi))
        left = r.wait_blocks(2)
        while len(inflight) > left:
            j, xj, yj = inflight.pop(0)
            assert (words(yj) == words(o.run_block(x[j * B:(j + 1) * B], C, C))).all(), f"block {j}"
            del xj, yj                                         # freed: the next np.copy may land on the same pages
    assert r.wait_blocks(0) == 0
    for j, xj, yj in inflight:
        assert (words(yj) == words(o.run_block(x[j * B:(j + 1) * B], C, C))).all(), f"block {j}"
    assert (r.sync_state() == o.state).all()


def test_two_chain_cores_through_the_queue():
    """two chain cores store into the same output window: queued one behind the other, neither loses its columns"""
    fmt, C, B = 6, 8, 512
    pw = pb.ProgramWriter(fmt)
    for half in range(2):
        pw.core()
        for c in range(half * C // 2, (half + 1) * C // 2):
            pw.param()
            bank = pw.biquad_bank(pb.synth_sections(c, 2, pb.F48000, pb.F48000))
            imp = pw.fir_impulses([pb.lcg_taps(c, 40)])
            pw.load_gain_fixed(C + c, 1.0); pw.biquads(bank, 2); pw.fir(imp, 40); pw.sat0db(); pw.store(c)
    prog = pw.end_of_code()
    x = pb.lcg_input(B * 4, C, True, seed=13)
    o = po.OracleProgram(fmt, prog)
    r = rt.Runtime(fmt, prog)
    assert len(r.cores) == 2
    outs = []
    for k in range(4):
        xi = np.ascontiguousarray(x[k * B:(k + 1) * B]); yi = np.full((B, C), 5.0, dtype=x.dtype)
        r.submit_block(xi, yi, C, 0)
        outs.append((xi, yi))
    assert r.wait_blocks(0) == 0
    for k, (xi, yi) in enumerate(outs):
        assert (words(yi) == words(o.run_block(x[k * B:(k + 1) * B], C, C))).all(), f"block {k}"
    assert (r.sync_state() == o.state).all()
