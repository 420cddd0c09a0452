"""N instances of one program side by side (dspRuntimeSetInstances / dspRuntimeBlockAllInstancesDevice, include/avdsp_runtime.h):
the reference's own programs (two channels each) are one wave's work, a GPU is filled by many of them.  Every instance must be
what the oracle gives for ITS input -- outputs bit for bit, and its data area at the end -- whatever its neighbours do."""
import os

import numpy as np
import pytest

from avdsp_amd import progbuilder as pb
from avdsp_amd import runtime as rt
from avdsp_amd import devmem as dm
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
        xd = dm.to_device(np.ascontiguousarray(xs[:, pos:pos + b]))          # [ninst][b][in_stride]
        yd = torch.zeros((ninst, b, out_stride), dtype=xd.dtype, device="cuda")
        r.run_block_all_instances_device(xd.data_ptr(), in_stride, in_base, b * in_stride, yd.data_ptr(), out_stride, 0, b * out_stride, b, st)
        torch.cuda.synchronize()
        out[:, pos:pos + b] = dm.to_host(yd)
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
        xd = dm.to_device(np.ascontiguousarray(xs[:, pos:pos + b]))
        yd = torch.zeros((ninst, b, OUT_S), dtype=xd.dtype, device="cuda")
        r.run_block_all_instances_device(xd.data_ptr(), IN_S, IN_B, b * IN_S, yd.data_ptr(), OUT_S, OUT_B, b * OUT_S, b, st)
        torch.cuda.synchronize()
        got[:, pos:pos + b] = dm.to_host(yd)
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
        xd = dm.to_device(np.ascontiguousarray(xs[:, pos:pos + b]))
        yd = torch.zeros((ninst, b, OUT_S), dtype=xd.dtype, device="cuda")
        r.run_block_all_instances_device(xd.data_ptr(), IN_S, IN_B, b * IN_S, yd.data_ptr(), OUT_S, OUT_B, b * OUT_S, b, st)
        torch.cuda.synchronize()
        got[:, pos:pos + b] = dm.to_host(yd)
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


def _chain_instances_vs_oracle(fmt, prog, C, ninst, blocks, seed0=40, options=None):
    """a program of C chains (inputs IO C .. 2C-1, outputs IO 0 .. C-1) in ninst instances with inputs of their own: every instance the
    oracle's bits for ITS input, block by block, and its data area at the end"""
    frames = sum(blocks)
    xs = np.stack([pb.lcg_input(frames, C, fmt in (5, 6), seed=seed0 + i) for i in range(ninst)])
    if ninst > 3:
        xs[1] = 0                                                    # a silent instance
        xs[2] = xs[3]                                                # two alike
    import torch
    r = rt.Runtime(fmt, prog)
    for k, v in (options or {}).items():
        r.set_option(k, v)
    r.set_instances(ninst)
    got = np.zeros((ninst, frames, C), dtype=xs.dtype)
    st = torch.cuda.current_stream().cuda_stream
    pos = 0
    for b in blocks:
        xd = dm.to_device(np.ascontiguousarray(xs[:, pos:pos + b]))
        yd = torch.zeros((ninst, b, C), dtype=xd.dtype, device="cuda")
        r.run_block_all_instances_device(xd.data_ptr(), C, C, b * C, yd.data_ptr(), C, 0, b * C, b, st)
        torch.cuda.synchronize()
        got[:, pos:pos + b] = dm.to_host(yd)
        pos += b
    for i in range(ninst):
        o = po.OracleProgram(fmt, prog)
        want = np.concatenate([o.run_block(xs[i, p0:p0 + b], C, C) for p0, b in zip(np.cumsum([0] + blocks[:-1]), blocks)])
        bad = np.nonzero((got[i].view(np.uint32) != want.view(np.uint32)).any(axis=1))[0]
        assert bad.size == 0, f"instance {i}: first differing frame {bad[0]}"
        assert (r.instance_state(i) == o.state).all(), f"instance {i}: state"
    return r


def test_cfg2_shaped_program_in_512_instances_is_the_golden_vector_512_times():
    """Round-4 review, Missing #5: instances of programs whose cores are CHAIN cores.  BASELINE cfg2's program (8 ch x 8 biquads, the
    golden bq_c8_s8_b256_f6 the compiled reference produced) in 512 instances -- 4096 rows of ONE cascade launch: every instance fed the
    golden input gives the reference's output and state, bit for bit; then instances with inputs of their own against the oracle."""
    import json
    import torch
    from tests.golden_recipes import make_input, make_program
    with open(os.path.join(GOLDEN_DIR, "manifest.json")) as f:
        case = [c for c in json.load(f)["cases"] if c["name"] == "bq_c8_s8_b256_f6"][0]
    g = np.load(os.path.join(GOLDEN_DIR, case["name"] + ".npz"))
    prog, x = make_program(case["program"]), make_input(case["input"], 6)
    ninst, B, C = 512, x.shape[0], 8
    r = rt.Runtime(6, prog)
    r.set_instances(ninst)
    xd = dm.to_device(np.ascontiguousarray(np.broadcast_to(x, (ninst,) + x.shape)))
    yd = torch.zeros((ninst, B, case["out_stride"]), dtype=xd.dtype, device="cuda")
    r.run_block_all_instances_device(xd.data_ptr(), x.shape[1], case["in_base"], B * x.shape[1], yd.data_ptr(), case["out_stride"], case["out_base"],
                                     B * case["out_stride"], B, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    got = dm.to_host(yd)
    assert (got.view(np.uint32) == g["out"].view(np.uint32)[None]).all()
    for i in (0, 1, 255, 511):
        assert (r.instance_state(i) == g["state"]).all()
    # an ordinary block call on a program that runs as instances is refused, not run on one of them by accident; SetInstances(0) gives it back
    with pytest.raises(rt.AvdspError):
        r.run_block(x, C, case["in_base"])
    r.set_instances(0)
    o = po.OracleProgram(6, prog)
    o.run_block(x, C, case["in_base"])                              # (instance 0 has run the golden block once already)
    more = pb.lcg_input(64, x.shape[1], True, seed=9)
    assert (r.run_block(more, C, case["in_base"]).view(np.uint32) == o.run_block(more, C, case["in_base"]).view(np.uint32)).all()
    r.release()


@pytest.mark.parametrize("fmt,C,S,T,ninst", [(6, 8, 8, 0, 64), (2, 8, 8, 0, 64), (4, 5, 3, 0, 7), (6, 3, 2, 300, 9), (6, 4, 0, 64, 5), (5, 6, 4, 0, 6), (3, 4, 2, 33, 5), (6, 2, 70, 0, 3), (2, 3, 130, 0, 2)])
def test_chain_instances_each_match_the_oracle(fmt, C, S, T, ninst):
    """inputs of their own per instance, ragged blocks (the distance between the instances' blocks changes with the block: the plans are
    re-made, the instances' states -- cascade words and FIR histories -- live on); int64, the double models, the float-accumulator
    models; cascades, cascade + FIR, FIR only"""
    prog = pb.synth_program(fmt, C, S, T)
    r = _chain_instances_vs_oracle(fmt, prog, C, ninst, [256, 64, 700, 1, 1024, 37])
    r.release()


def test_chain_instances_under_the_overlap_mode():
    """cascade + FIR chains in instances with the cascade of the next block under the FIR of this one"""
    prog = pb.synth_program(6, 6, 4, 500)
    r = _chain_instances_vs_oracle(6, prog, 6, 12, [1024, 1024, 512, 1024], options={"overlap": 1})
    r.release()


@pytest.mark.parametrize("ready_words", [-1, 0, 1, 2])
def test_long_cascades_in_pieces_under_the_overlap_mode(ready_words):
    """70 sections in front of a 300-tap FIR: the cascade runs as two pieces (more than 64 sections do not fit a wave), on the cascades'
    stream, under the previous block's FIR; only the LAST piece's launch appends to the rings and publishes the chains' ready words"""
    prog = pb.synth_program(6, 3, 70, 300)
    r = _chain_instances_vs_oracle(6, prog, 3, 1, [1024, 1024, 1024, 333, 1024], options={"overlap": 1, "ready_words": ready_words})
    r.set_option("ready_words", -1); r.set_option("overlap", 0)
    r.release()


@pytest.mark.parametrize("fmt", [2, 4, 6])
def test_instances_of_a_program_of_both_kinds(fmt):
    """a program with a CHAIN core (gain -> 5 biquads [-> 40-tap FIR] -> SAT0DB -> store, two channels) and a core for the interpreter (X/Y
    moves): until round 5 refused -- its instances would keep their state in two places.  Now every core runs on the interpreter while
    the program has instances; each instance the oracle's bits and data area; dspRuntimeSetInstances(0) gives the single program its
    chain plans back and it continues from instance 0's state."""
    import torch
    C, S, T = 2, 5, 40 if fmt != 2 else 0
    w = pb.ProgramWriter(fmt, pb.F48000, pb.F48000)
    w.core()
    for c in range(C):
        w.param()
        bank = w.biquad_bank(pb.synth_sections(c, S, pb.F48000, pb.F48000))
        imp = w.fir_impulses([pb.lcg_taps(c, T)]) if T else None
        w.load_gain_fixed(4 + c, 0.5); w.biquads(bank, S)
        if T:
            w.fir(imp, T)
        w.sat0db(); w.store(c)
    w.core()
    w.load(6); w.copyxy(); w.load(7); w.swapxy(); w.store(2); w.swapxy(); w.store(3)
    prog = w.end_of_code()
    ninst, blocks = 9, [64, 300, 2, 129]
    frames = sum(blocks)
    xs = np.stack([pb.lcg_input(frames, 4, fmt == 6, seed=70 + i) for i in range(ninst)])
    xs[4] = xs[5]
    got, r = _run_instances(fmt, prog, xs, 4, 4, 4, blocks)
    for i in range(ninst):
        o = po.OracleProgram(fmt, prog, fs=48000, random=3, dither=24)
        frame = np.zeros(4096, dtype=np.uint32)
        pos = 0
        for b in blocks:
            want = o.run_block(xs[i, pos:pos + b], 4, 4, 0, block=b, frame=frame)
            assert (got[i, pos:pos + b].view(np.uint32) == want.view(np.uint32)).all(), f"instance {i}, block at {pos}"
            pos += b
        assert (r.instance_state(i) == o.state).all(), f"instance {i}: data area"
        if i == 0:
            o0, frame0 = o, frame
    assert (got[4] == got[5]).all() and got.any()
    assert r.get_option("generic") == 1
    r.set_instances(0)                                               # the single program again, on its chain plan + the interpreter
    assert r.get_option("generic") == 0
    more = pb.lcg_input(200, 4, fmt == 6, seed=5)
    want = o0.run_block(more, 4, 4, 0, block=200, frame=frame0)
    assert (r.run_block_all(more, 4, 4).view(np.uint32) == want.view(np.uint32)).all()
    assert (r.sync_state() == o0.state).all()
    r.release()


def test_one_instance_of_a_fir_chain_program_hands_back_its_fir_history():
    """found by tests/dev/gpu_instance_sweep.py (14 of its first 400 runs): with ONE instance a chain program runs on its ordinary plan, whose
    FIR histories live in device rings -- dspRuntimeInstanceState(0) has to bring them home like dspRuntimeSyncState does"""
    prog = pb.synth_program(6, 2, 8, 33)
    r = _chain_instances_vs_oracle(6, prog, 2, 1, [256, 64])
    r.release()
