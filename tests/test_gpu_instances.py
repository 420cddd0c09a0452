"""N instances of one program side by side (dspRuntimeSetInstances / dspRuntimeBlockAllInstancesDevice, include/avdsp_runtime.h):
the reference's own programs (two channels each) are one wave's work, a GPU is filled by many of them.  Every instance must be
what the oracle gives for ITS input -- outputs bit for bit, and its data area at the end -- whatever its neighbours do."""
import os

import numpy as np
import pytest

from avdsp_amd import progbuilder as pb
from avdsp_amd import runtime as rt
from oracle import pyoracle as po
from tests.golden_recipes import GOLDEN_DIR

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _release():
    yield
    rt.lib().dspRuntimeSetOption(b"strand_lanes", 1)
    rt.lib().dspRuntimeRelease()


def _run_instances(fmt, prog, xs, in_stride, in_base, out_stride, blocks, fs=48000, seed=3, dither=24):
    """xs: [ninst][frames][in_stride]; returns ([ninst][frames][out_stride], runtime)"""
    import torch
    ninst, frames = xs.shape[0], xs.shape[1]
    r = rt.Runtime(fmt, prog, fs=fs, random=seed, dither=dither)
    assert r.rc > 0
    r.set_instances(ninst)
    out = np.zeros((ninst, frames, out_stride), dtype=xs.dtype)
    st = torch.cuda.current_stream().cuda_stream
    pos = 0
    for b in blocks:
        xd = torch.from_numpy(np.ascontiguousarray(xs[:, pos:pos + b])).cuda()          # [ninst][b][in_stride]
        yd = torch.zeros((ninst, b, out_stride), dtype=xd.dtype, device="cuda")
        r.run_block_all_instances_device(xd.data_ptr(), in_stride, in_base, b * in_stride, yd.data_ptr(), out_stride, 0, b * out_stride, b, st)
        torch.cuda.synchronize()
        out[:, pos:pos + b] = yd.cpu().numpy()
        pos += b
    return out, r


def test_instances_of_the_reference_crossover_each_match_the_oracle():
    """crossoverLV6.bin (dspprogs/crossoverLV6.c: inputs IO 16, 17; outputs IO 25 .. 29) in 37 instances with inputs of their own.
    (windows that share no IO number: the pieces of a level go out as one grid; the other way round: the next test)"""
    prog = np.fromfile(os.path.join(GOLDEN_DIR, "crossoverLV6.bin"), dtype=np.uint32)
    ninst, blocks = 37, [64, 300, 2, 129]
    frames = sum(blocks)
    IN_S, IN_B, OUT_S, OUT_B = 2, 16, 8, 24
    xs = np.stack([pb.lcg_input(frames, IN_S, False, seed=100 + i) for i in range(ninst)])
    xs[5] = 0                                                        # a silent instance among the others
    xs[6] = xs[7]                                                    # two with the same input: the same output
    import torch
    r = rt.Runtime(2, prog, fs=48000, random=3, dither=24)
    r.set_instances(ninst)
    got = np.zeros((ninst, frames, OUT_S), dtype=xs.dtype)
    st = torch.cuda.current_stream().cuda_stream
    pos = 0
    for b in blocks:
        xd = torch.from_numpy(np.ascontiguousarray(xs[:, pos:pos + b])).cuda()
        yd = torch.zeros((ninst, b, OUT_S), dtype=xd.dtype, device="cuda")
        r.run_block_all_instances_device(xd.data_ptr(), IN_S, IN_B, b * IN_S, yd.data_ptr(), OUT_S, OUT_B, b * OUT_S, b, st)
        torch.cuda.synchronize()
        got[:, pos:pos + b] = yd.cpu().numpy()
        pos += b
    for i in range(ninst):
        o = po.OracleProgram(2, prog, fs=48000, random=3, dither=24)
        frame = np.zeros(4096, dtype=np.uint32)
        pos = 0
        want = np.zeros((frames, OUT_S), dtype=xs.dtype)
        for b in blocks:
            want[pos:pos + b] = o.run_block(xs[i, pos:pos + b], OUT_S, IN_B, OUT_B, block=b, frame=frame)
            pos += b
        bad = np.nonzero((got[i].view(np.uint32) != want.view(np.uint32)).any(axis=0))[0]
        assert bad.size == 0, f"instance {i}: output columns {list(bad)} differ"
        assert (r.instance_state(i) == o.state).all(), f"instance {i}: data area differs"
    assert (got[6] == got[7]).all() and got[1:5].any()
    r.release()


@pytest.mark.parametrize("name", ["dacdiy1.bin", "crossoverLV6.bin"])
def test_instances_with_windows_that_share_io_numbers(name):
    """the goldens' windows (input IO 8 .. 23 inside output IO 0 .. 31): dacdiy1.bin's outputs lie on both sides of its inputs, so its
    windows cannot be kept apart.  Whole rows then move (the input shows through where the program stores nothing) and the pieces
    of a level run one after the other, each as a grid over the instances; every instance is still the oracle's bits."""
    import torch
    prog = np.fromfile(os.path.join(GOLDEN_DIR, name), dtype=np.uint32)
    ninst, blocks = 11, [64, 200, 3, 130]
    frames = sum(blocks)
    IN_S, IN_B, OUT_S, OUT_B = 16, 8, 32, 0
    xs = np.stack([pb.lcg_input(frames, IN_S, False, seed=40 + i) for i in range(ninst)])
    xs[2] = xs[9]
    r = rt.Runtime(2, prog, fs=48000, random=3, dither=24)
    r.set_instances(ninst)
    got = np.zeros((ninst, frames, OUT_S), dtype=xs.dtype)
    st = torch.cuda.current_stream().cuda_stream
    pos = 0
    for b in blocks:
        xd = torch.from_numpy(np.ascontiguousarray(xs[:, pos:pos + b])).cuda()
        yd = torch.zeros((ninst, b, OUT_S), dtype=xd.dtype, device="cuda")
        r.run_block_all_instances_device(xd.data_ptr(), IN_S, IN_B, b * IN_S, yd.data_ptr(), OUT_S, OUT_B, b * OUT_S, b, st)
        torch.cuda.synchronize()
        got[:, pos:pos + b] = yd.cpu().numpy()
        pos += b
    for i in range(ninst):
        o = po.OracleProgram(2, prog, fs=48000, random=3, dither=24)
        frame = np.zeros(4096, dtype=np.uint32)
        pos = 0
        for b in blocks:
            want = o.run_block(xs[i, pos:pos + b], OUT_S, IN_B, OUT_B, block=b, frame=frame)
            bad = np.nonzero((got[i, pos:pos + b].view(np.uint32) != want.view(np.uint32)).any(axis=0))[0]
            assert bad.size == 0, f"{name} instance {i}, block at {pos}: output columns {list(bad)} differ"
            pos += b
        assert (r.instance_state(i) == o.state).all(), f"{name} instance {i}: data area differs"
    assert (got[2] == got[9]).all()
    r.release()


@pytest.mark.parametrize("fmt", [2, 5, 6])
def test_instances_of_a_random_program(fmt):
    """a random multi-core program (memories, dither, delay lines ...) in 20 instances, float models included"""
    from tests.fuzz_programs import IN_BASE, N_IN, N_OUT, random_program
    prog = random_program(112, fmt)
    ninst, blocks = 20, [100, 64, 7]
    frames = sum(blocks)
    xs = np.stack([pb.lcg_input(frames, N_IN, fmt in (5, 6), seed=7 + i) for i in range(ninst)])
    got, r = _run_instances(fmt, prog, xs, N_IN, IN_BASE, N_OUT, blocks, seed=112)
    for i in range(ninst):
        o = po.OracleProgram(fmt, prog, fs=48000, random=112, dither=24)
        frame = np.zeros(4096, dtype=np.uint32)
        pos = 0
        for b in blocks:
            want = o.run_block(xs[i, pos:pos + b], N_OUT, IN_BASE, 0, block=b, frame=frame)
            assert (got[i, pos:pos + b].view(np.uint32) == want.view(np.uint32)).all(), f"instance {i}, block at {pos}"
            pos += b
        assert (r.instance_state(i) == o.state).all()
    r.release()


def test_instances_refuse_what_they_cannot_run():
    """a chain program (the parallel kernels know nothing of instances) is refused loudly, not run wrongly"""
    import torch
    prog = pb.synth_program(6, 4, 2, 0)
    r = rt.Runtime(6, prog)
    r.set_instances(3)
    x = torch.zeros((3, 64, 4), dtype=torch.float32, device="cuda")
    y = torch.zeros((3, 64, 4), dtype=torch.float32, device="cuda")
    with pytest.raises(rt.AvdspError):
        r.run_block_all_instances_device(x.data_ptr(), 4, 4, 64 * 4, y.data_ptr(), 4, 0, 64 * 4, 64, 0)
    r.release()
