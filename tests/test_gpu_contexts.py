"""Several programs in one process (contexts keyed by the caller's buffer): interleaved calls give what each program
gives alone; one process drives "two GPUs" (two buffers of the same program, one shard each; the test box has one GPU, both
contexts sit on it); releasing one program leaves the other intact."""
import ctypes as C

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


def test_two_programs_interleaved():
    pa = pb.synth_program(6, 10, 3, 200)            # cascade + FIR, double model
    pbq = pb.synth_program(2, 7, 5, 0)              # cascade, int64 model
    xa = pb.lcg_input(4 * 512, 10, True, seed=1)
    xb = pb.lcg_input(4 * 512, 7, False, seed=2)
    oa, ob = po.OracleProgram(6, pa), po.OracleProgram(2, pbq)
    ra = rt.Runtime(6, pa)
    rb = rt.Runtime(2, pbq)
    hdr = C.c_void_p.in_dll(ra.L, "dspHeaderPtr")
    for k in range(4):
        sl = slice(k * 512, (k + 1) * 512)
        ga = ra.run_block(xa[sl], 10, 10)
        assert hdr.value == ra.buf.ctypes.data        # the reference's exported data follow the program of the latest call
        gb = rb.run_block(xb[sl], 7, 7)
        assert hdr.value == rb.buf.ctypes.data
        assert (words(ga) == words(oa.run_block(xa[sl], 10, 10))).all(), f"program A block {k}"
        assert (words(gb) == words(ob.run_block(xb[sl], 7, 7))).all(), f"program B block {k}"
    assert (ra.sync_state() == oa.state).all()
    assert (rb.sync_state() == ob.state).all()
    # options and shard are per program (and defaults for programs loaded later)
    ra.set_shard(1, 2)
    assert ra.get_option("shard_rank") == 1 and rb.get_option("shard_rank") == 0
    ra.set_shard(0, 1)


def test_one_process_two_shards_of_one_program():
    fmt, Cn, S, T, B = 6, 12, 2, 150, 1024
    prog = pb.synth_program(fmt, Cn, S, T)
    x = pb.lcg_input(3 * B, Cn, True, seed=7)
    o = po.OracleProgram(fmt, prog)
    ranks = [rt.Runtime(fmt, prog.copy()) for _ in range(2)]           # one buffer per "GPU"
    infos = []
    for r, rk in zip(ranks, (0, 1)):
        r.set_shard(rk, 2)
        infos.append(r.shard_info())
    assert [i["nchains"] for i in infos] == [6, 6] and infos[1]["first_chain"] == 6
    for k in range(3):
        want = o.run_block(x[k * B:(k + 1) * B], Cn, Cn)
        got = np.zeros_like(want)
        for r, i in zip(ranks, infos):
            lo = i["in_io_min"] - Cn
            xs = np.ascontiguousarray(x[k * B:(k + 1) * B, lo:lo + i["nchains"]])
            got[:, i["out_io_min"]:i["out_io_min"] + i["nchains"]] = r.run_block(xs, i["nchains"], i["in_io_min"], i["out_io_min"])
        assert (words(got) == words(want)).all(), f"block {k}"
    for r in ranks:
        r.set_shard(0, 1)


def test_release_one_program_and_load_a_buffer_again():
    p1 = pb.synth_program(6, 4, 2, 60)
    p2 = pb.synth_program(6, 5, 1, 0)
    x1 = pb.lcg_input(1024, 4, True, seed=3)
    x2 = pb.lcg_input(1024, 5, True, seed=4)
    r1, r2 = rt.Runtime(6, p1), rt.Runtime(6, p2)
    r1.run_block(x1[:512], 4, 4)
    r2.run_block(x2[:512], 5, 5)
    r1.release()
    with pytest.raises(rt.AvdspError):
        r1.run_block(x1[512:], 4, 4)                                   # its context is gone (rundata no longer belongs to a program)
    o2 = po.OracleProgram(6, p2)
    o2.run_block(x2[:512], 5, 5)
    assert (words(r2.run_block(x2[512:], 5, 5)) == words(o2.run_block(x2[512:], 5, 5))).all()
    # the same buffer loaded again: a clean start
    buf = r2.buf
    L = r2.L
    assert L.dspRuntimeInit(buf.ctypes.data, len(buf), 48000, 0, 31) > 0
    o3 = po.OracleProgram(6, p2)
    assert (words(r2.run_block(x2[:512], 5, 5)) == words(o3.run_block(x2[:512], 5, 5))).all()
