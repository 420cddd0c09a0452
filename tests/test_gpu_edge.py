"""Edge cases of the HIP path against the oracle: long and odd tap counts (several LDS chunks), deep
cascades (lanes-per-channel 32, 64 and the > 64 fallback), chains of different shape in one core,
several cores, IO windows with offsets and spare slots, blocks longer than one launch, rate changes,
reset, and the asynchronous device entry point on a side stream."""
import os

import numpy as np
import pytest

from avdsp_amd import progbuilder as pb
from avdsp_amd import runtime as rt
from avdsp_amd import devmem as dm
from oracle import pyoracle as po
from tests.test_gpu_parity import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _release():
    yield
    rt.lib().dspRuntimeRelease()


def both(fmt, prog, x, out_stride, in_base, out_base=0, blocks=None, opts=None, fs=48000, scratch=None):
    o = po.OracleProgram(fmt, prog, fs=fs)
    r = rt.Runtime(fmt, prog, fs=fs)
    for k, v in (opts or {}).items():
        r.set_option(k, v)
    blocks = blocks or [len(x)]
    pos = 0
    for b in blocks:
        want = o.run_block(x[pos:pos + b], out_stride, in_base, out_base, scratch_len=scratch)
        got = r.run_block(x[pos:pos + b], out_stride, in_base, out_base)
        assert_close(got, want, fmt, what=f"frames {pos}..{pos + b}")
        pos += b
    assert (r.sync_state() == o.state).all()
    return r, o


@pytest.mark.parametrize("taps,fir_impl", [(1249, 1), (1249, 0), (2500, 1), (5000, 1), (9001, 1), (20000, 1)])
def test_long_fir_many_chunks(taps, fir_impl):
    prog = pb.synth_program(6, 2, 1, taps)
    blocks = [300, 1024, 3, 500]
    x = pb.lcg_input(sum(blocks), 2, True, seed=taps)
    both(6, prog, x, 2, 2, blocks=blocks, opts={"fir_impl": fir_impl})


@pytest.mark.parametrize("fmt", [2, 6])
@pytest.mark.parametrize("sections", [17, 31, 33, 64, 65, 100])
def test_deep_cascades(fmt, sections):
    prog = pb.synth_program(fmt, 3, sections)
    x = pb.lcg_input(150, 3, fmt == 6, seed=sections)
    both(fmt, prog, x, 3, 3, blocks=[70, 1, 79])


def test_mixed_shapes_in_one_core():
    """chain 0: 2 sections + 31 taps; chain 1: 5 sections; chain 2: FIR only, 100 taps; chain 3: plain copy;
    chain 4: 5 sections + 300 taps.  IO: inputs 40..44, outputs 20, 3, 7, 0, 30 (windows do not overlap)."""
    fmt = 6
    pw = pb.ProgramWriter(fmt)
    pw.core()
    pw.param()
    b0 = pw.biquad_bank(pb.synth_sections(0, 2, pb.F48000, pb.F48000))
    i0 = pw.fir_impulses([pb.lcg_taps(0, 31)])
    b1 = pw.biquad_bank(pb.synth_sections(1, 5, pb.F48000, pb.F48000))
    pw.param()
    i2 = pw.fir_impulses([pb.lcg_taps(2, 100)])
    b4 = pw.biquad_bank(pb.synth_sections(4, 5, pb.F48000, pb.F48000))
    i4 = pw.fir_impulses([pb.lcg_taps(4, 300)])
    pw.load_gain_fixed(40, 0.9); pw.biquads(b0, 2); pw.fir(i0, 31); pw.sat0db(); pw.store(20)
    pw.load_gain_fixed(41, 1.1); pw.biquads(b1, 5); pw.store(3)
    pw.load(42); pw.fir(i2, 100); pw.sat0db(); pw.store(7)
    pw.load(43); pw.store(0)
    pw.load_gain_fixed(44, 0.7); pw.biquads(b4, 5); pw.fir(i4, 300); pw.store(30)
    prog = pw.end_of_code()
    x = pb.lcg_input(900, 5, True, seed=21)
    r, _ = both(fmt, prog, x, 31, 40, 0, blocks=[400, 500], scratch=48)
    assert r.core_info() == dict(chains=5, max_sections=5, max_taps=300)


def test_two_cores_and_untouched_slots():
    fmt = 2
    pw = pb.ProgramWriter(fmt)
    pw.core()
    pw.param()
    b0 = pw.biquad_bank(pb.synth_sections(0, 3, pb.F48000, pb.F48000))
    pw.load_gain_fixed(8, 1.0); pw.biquads(b0, 3); pw.sat0db(); pw.store(0)
    pw.core()
    pw.param()
    b1 = pw.biquad_bank(pb.synth_sections(1, 2, pb.F48000, pb.F48000))
    pw.load_gain_fixed(9, 0.5); pw.biquads(b1, 2); pw.sat0db(); pw.store(2)
    prog = pw.end_of_code()
    x = pb.lcg_input(300, 4, False, seed=5)                 # IO 8..11 offered, 10 and 11 unused
    o = po.OracleProgram(fmt, prog)
    r = rt.Runtime(fmt, prog)
    assert len(r.cores) == 2
    out_o = np.full((300, 4), 77, dtype=np.int32)
    out_d = out_o.copy()
    o.run_block(x, 4, 8, 0, out=out_o)
    r.run_block(x, 4, 8, 0, out=out_d)
    assert (out_o == out_d).all()
    assert (out_d[:, 1] == 77).all() and (out_d[:, 3] == 77).all()      # never stored: left as they were
    assert (r.sync_state() == o.state).all()


@pytest.mark.parametrize("fmt", [2, 6])
def test_block_longer_than_one_launch(fmt):
    prog = pb.synth_program(fmt, 4, 3, 0 if fmt == 2 else 70)
    x = pb.lcg_input(2500, 4, fmt == 6, seed=9)
    both(fmt, prog, x, 4, 4, blocks=[2500])


@pytest.mark.parametrize("fmt", [2, 6])
def test_rate_change_and_reset(fmt):
    """Multi-rate program (44.1k..96k encoded): dspRuntimeReset picks another coefficient column and
    zeroes the state; a second run after Reset equals a fresh run."""
    prog = pb.synth_program(fmt, 3, 4, 0, pb.F44100, pb.F96000)
    x = pb.lcg_input(200, 3, fmt == 6, seed=2)
    r = rt.Runtime(fmt, prog, fs=44100)
    o = po.OracleProgram(fmt, prog, fs=44100)
    assert_close(r.run_block(x, 3, 3), o.run_block(x, 3, 3), fmt)
    for fs in (96000, 48000):
        assert r.reset(fs) == 0 and o.reset(fs) == 0
        assert_close(r.run_block(x, 3, 3), o.run_block(x, 3, 3), fmt, what=f"fs {fs}")
        assert (r.sync_state() == o.state).all()
    assert r.reset(192000) == -2                           # outside the encoded range


def test_device_entry_point_on_a_side_stream():
    """dspRuntimeBlockDevice only enqueues kernels on the stream it is given (no sync, no allocation
    after the plan exists): blocks issued on a non-default stream, inputs refilled in stream order."""
    torch = pytest.importorskip("torch")
    C, S, T, B = 16, 4, 200, 256
    prog = pb.synth_program(6, C, S, T)
    x = pb.lcg_input(3 * B, C, True, seed=14)
    o = po.OracleProgram(6, prog)
    want = o.run_block(x, C, C)
    r = rt.Runtime(6, prog)
    xd = torch.zeros((B, C), dtype=torch.float32, device="cuda")
    yd = torch.zeros((B, C), dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        xd.copy_(torch.from_numpy(x[:B]))
        r.run_block_device(xd.data_ptr(), C, C, yd.data_ptr(), C, 0, B, side.cuda_stream)     # plan creation outside capture
    side.synchronize()
    got = [dm.to_host(yd)]
    for k in (1, 2):
        with torch.cuda.stream(side):
            xd.copy_(torch.from_numpy(x[k * B:(k + 1) * B]))
            r.run_block_device(xd.data_ptr(), C, C, yd.data_ptr(), C, 0, B, side.cuda_stream)
        side.synchronize()
        got.append(dm.to_host(yd))
    assert_close(np.concatenate(got), want, 6)


# ---------------------------------------------------------------------------------------------
# host sample formats in front of the block (linux/avdsp_plugin.c:103-121)
# ---------------------------------------------------------------------------------------------
def _pack(samples32: np.ndarray, pcm: int) -> np.ndarray:
    """int32 s.31 words -> packed little-endian PCM bytes, dropping the low bits the format lacks."""
    u = samples32.astype(np.int32).view(np.uint32).reshape(-1)
    if pcm == rt.PCM_S16:
        return (u >> 16).astype("<u2").view(np.uint8)
    b = np.empty((u.size, 3), dtype=np.uint8)
    b[:, 0] = (u >> 8) & 0xFF; b[:, 1] = (u >> 16) & 0xFF; b[:, 2] = (u >> 24) & 0xFF
    return b.reshape(-1)


def _unpack_like_the_plugin(raw: np.ndarray, pcm: int) -> np.ndarray:
    if pcm == rt.PCM_S16:
        return (raw.view("<i2").astype(np.int32) << 16)
    b = raw.reshape(-1, 3).astype(np.uint32)
    return ((b[:, 0] << 8) | (b[:, 1] << 16) | (b[:, 2] << 24)).view(np.int32)


@pytest.mark.parametrize("pcm", [rt.PCM_S16, rt.PCM_S24_3LE, rt.PCM_S32])
@pytest.mark.parametrize("fmt,channels,frames", [(2, 6, 333), (4, 5, 127), (2, 1, 1)])
def test_packed_pcm_blocks(pcm, fmt, channels, frames):
    """S16 / S24_3LE / S32 host buffers through dspRuntimeBlockPcm == the plugin's unpacking followed
    by the block; ragged sizes exercise the sample-by-sample tail of the unpack kernel."""
    prog = pb.synth_program(fmt, channels, 3, 0 if fmt == 2 else 9)
    x = pb.lcg_input(frames, channels, False, seed=21)
    raw = x.view(np.uint8).reshape(-1) if pcm == rt.PCM_S32 else _pack(x, pcm)
    xq = x if pcm == rt.PCM_S32 else _unpack_like_the_plugin(raw, pcm).reshape(frames, channels)
    o = po.OracleProgram(fmt, prog)
    want = o.run_block(xq, channels, channels)
    r = rt.Runtime(fmt, prog)
    got = r.run_block_pcm(pcm, raw, frames, channels, channels, channels)
    assert_close(got, want, fmt)
    assert (r.sync_state() == o.state).all()


def test_packed_pcm_into_the_general_interpreter_and_refusal():
    prog = np.fromfile(os.path.join(os.path.dirname(__file__), "golden", "crossoverLV6.bin"), dtype=np.uint32)
    x = pb.lcg_input(200, 16, False, seed=3)
    raw = _pack(x, rt.PCM_S24_3LE)
    xq = _unpack_like_the_plugin(raw, rt.PCM_S24_3LE).reshape(200, 16)
    o = po.OracleProgram(2, prog, fs=48000, random=9, dither=24)
    want = o.run_block(xq, 8, 16, 24, scratch_len=40, block=50)      # inputs IO 16..31, outputs IO 24..31 -> separate windows
    r = rt.Runtime(2, prog, fs=48000, random=9, dither=24)
    got = r.run_block_pcm(rt.PCM_S24_3LE, raw, 200, 16, 8, 16, 24, block=50)
    # the plugin reads inputs at IO 16.. and writes outputs from IO 24..: the windows overlap in IO numbers, which the
    # host loop (and the oracle's) resolves by laying the inputs over the frame first
    assert (got == want).all()
    r6 = rt.Runtime(6, pb.synth_program(6, 2, 1))
    with pytest.raises(rt.AvdspError) as e:
        r6._check(r6.L.dspRuntimeBlockPcm(6, r6.cores[0], r6.rundata, rt.PCM_S16, raw.ctypes.data, 2, 2,
                                          np.zeros(8, dtype=np.int32).ctypes.data, 2, 0, 2))
    assert e.value.code == -1 and "int-sample" in str(e.value)


@pytest.mark.parametrize("pcm", [rt.PCM_S16, rt.PCM_S24_3LE, rt.PCM_S32])
def test_packed_pcm_through_every_core_at_once(pcm):
    """dspRuntimeBlockAllPcm = the plugin's whole transfer function: unpack once, every core, S32 out"""
    prog = np.fromfile(os.path.join(os.path.dirname(__file__), "golden", "dacdiy1.bin"), dtype=np.uint32)
    x = pb.lcg_input(300, 8, False, seed=4)
    if pcm == rt.PCM_S32:
        raw, xq = np.ascontiguousarray(x).view(np.uint8).reshape(-1), x
    else:
        raw = _pack(x, pcm)
        xq = _unpack_like_the_plugin(raw, pcm).reshape(300, 8)
    o = po.OracleProgram(2, prog, fs=48000, random=2, dither=24)
    want = o.run_block(xq, 8, 8, 0, block=100, frame=np.zeros(4096, dtype=np.uint32))     # the plugin's IO layout: out 0..7, in 8..15
    r = rt.Runtime(2, prog, fs=48000, random=2, dither=24)
    got = r.run_block_all_pcm(pcm, raw, 300, 8, 8, 8, 0, block=100)
    assert (got == want).all()
    assert (r.sync_state() == o.state).all()
    assert (r.get_option("levels"), r.get_option("cores")) == (2, 4)
    r.release()


def test_unpack_on_device_misaligned_source():
    torch = pytest.importorskip("torch")
    r = rt.Runtime(2, pb.synth_program(2, 2, 1))
    r.run_block(np.zeros((1, 2), dtype=np.int32), 2, 2)                  # creates the device program
    x = pb.lcg_input(1001, 3, False, seed=8)
    for pcm in (rt.PCM_S16, rt.PCM_S24_3LE):
        raw = _pack(x, pcm)
        for shift in (0, 1, 2):                                          # source start not dword aligned
            buf = torch.zeros(raw.size + 8, dtype=torch.uint8, device="cuda")
            buf[shift:shift + raw.size] = torch.from_numpy(raw.copy())
            dst = torch.zeros(x.size + 4, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            rc = r.L.dspRuntimeUnpackPcmDevice(pcm, buf.data_ptr() + shift, dst.data_ptr(), x.size, None)
            assert rc == 0, r.last_error()
            torch.cuda.synchronize()
            got = dm.to_host(dst)
            assert (got[:x.size] == _unpack_like_the_plugin(raw, pcm)).all()
            assert (got[x.size:] == 0).all()


# ---------------------------------------------------------------------------------------------
# a C host linked against the library (the drop-in boundary used from C, not through ctypes)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fmt,which", [(2, "crossoverLV6"), (6, "synth"), (4, "synth")])
def test_c_host_links_and_matches_the_oracle(tmp_path, fmt, which):
    """examples/host_demo.c uses the reference's API (dspRuntimeInit, dspFindCore, dspFindCoreBegin) plus
    dspRuntimeBlock_N, selected by -DDSP_FORMAT like the reference's hosts; built with gcc, linked with
    -lavdsp_mi355x, run as its own process."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "host_demo")
    subprocess.check_call(["gcc", "-std=gnu99", "-Wall", f"-I{root}/include", f"-DDSP_FORMAT={fmt}",
                           f"{root}/examples/host_demo.c", f"-L{root}/avdsp_amd/lib", "-lavdsp_mi355x",
                           f"-Wl,-rpath,{root}/avdsp_amd/lib", "-o", exe])
    if which == "crossoverLV6":
        prog = np.fromfile(os.path.join(root, "tests", "golden", "crossoverLV6.bin"), dtype=np.uint32)
        nin, in_base, nout, out_base = 8, 16, 8, 24
    else:
        prog = pb.synth_program(6, 3, 2, 17)
        nin, in_base, nout, out_base = 3, 3, 3, 0
    x = pb.lcg_input(700, nin, fmt == 6, seed=11)
    (tmp_path / "p.bin").write_bytes(prog.tobytes())
    (tmp_path / "in.raw").write_bytes(x.tobytes())
    res = subprocess.run([exe, str(tmp_path / "p.bin"), "48000", str(tmp_path / "in.raw"), str(nin), str(in_base),
                          str(tmp_path / "out.raw"), str(nout), str(out_base), "256"], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    got = np.fromfile(str(tmp_path / "out.raw"), dtype=x.dtype).reshape(700, nout)
    o = po.OracleProgram(fmt, prog, fs=48000, random=12345, dither=24)
    want = o.run_block(x, nout, in_base, out_base, scratch_len=40, block=256)
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    words = res.stdout.split("state[0..3]=")[1].split()
    assert [int(w, 16) for w in words[:4]] == [int(v) for v in o.state[:4]]


def test_inf_and_nan_samples_follow_the_reference_products():
    """dspMulFloatDouble builds its products from bit fields and reads exponent 255 as 1.m x 2^128
    (dsp_ieee754.h:377-410).  The FIR, the lane-per-channel cascade and the interpreter compute that way throughout; the
    pipelined cascade (the default) computes with IEEE values, notices at the end of a block that an Inf or NaN has been
    through a chain (non-finite accumulator or exponent-255 state) and runs that chain's block again in the reference's
    order from the untouched state."""
    prog = pb.synth_program(6, 2, 2, 5)
    x = np.zeros((24, 2), dtype=np.float32)
    x[0, 0] = np.inf; x[5, 0] = np.nan; x[9, 1] = -np.inf; x[12, 0] = 3e38; x[13, 1] = -3.4e38
    for opts in (dict(biquad_impl=1, fir_impl=1), dict(biquad_impl=0, fir_impl=1), dict(biquad_impl=0, fir_impl=0), dict(generic=1)):
        o = po.OracleProgram(6, prog)
        r = rt.Runtime(6, prog)
        for k, v in opts.items():
            r.set_option(k, v)
        try:
            want = o.run_block(x, 2, 2, 0, block=24)
            got = r.run_block(x, 2, 2, 0, block=24)
            assert (got.view(np.uint32) == want.view(np.uint32)).all(), opts
            assert (r.sync_state() == o.state).all(), opts
        finally:
            r.set_option("generic", 0)
            r.set_option("biquad_impl", 1)
            r.release()


@pytest.mark.parametrize("fmt", [4, 6])
@pytest.mark.parametrize("sections,taps", [(16, 0), (5, 40), (24, 0), (40, 7)])
def test_pipelined_cascade_replays_chains_that_met_inf_or_nan(fmt, sections, taps):
    """Some channels of a wide program get Inf, NaN, overflowing or exponent-255 input at odd places (first and last frame of a
    block, inside), the others ordinary noise; several blocks, so that exponent-255 STATE is carried into a block as well.
    Every channel must be the oracle's, bit for bit: the odd ones through the replay, the rest untouched by it.  Format 4
    reaches Inf only through overflow (int samples): a gain of 4 on a cascade that amplifies does it."""
    C = 37
    prog = pb.synth_program(fmt, C, sections, taps, gain=1.0)
    blocks = [64, 1, 300, 17, 200]
    n = sum(blocks)
    x = pb.lcg_input(n, C, fmt == 6, seed=77)
    if fmt == 6:
        xi = x.view(np.uint32)
        xi[0, 3] = 0x7F800000; xi[63, 5] = 0xFF800000; xi[64, 9] = 0x7FC00001; xi[100, 16] = 0x7F7FFFFF; xi[101, 16] = 0x7F7FFFFF
        xi[364, 20] = 0xFFFFFFFF; xi[365 + 16, 31] = 0x7F812345; xi[n - 1, 36] = 0x7F800000
        x[200:230, 12] = 3.0e38                                # overflows inside the cascade, not at its input
    o = po.OracleProgram(fmt, prog)
    r = rt.Runtime(fmt, prog)
    pos = 0
    for b in blocks:
        want = o.run_block(x[pos:pos + b], C, C)
        got = r.run_block(x[pos:pos + b], C, C)
        bad = np.nonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=0))[0]
        assert bad.size == 0, f"block at {pos}: channels {bad.tolist()} differ"
        pos += b
    assert (r.sync_state() == o.state).all()


@pytest.mark.parametrize("fmt,sections", [(6, 16), (6, 24), (5, 8), (3, 5)])
def test_in_place_blocks_with_inf_and_nan(fmt, sections):
    """A device-resident block processed where it lies (input and output windows in the same memory, different IO numbers) with
    Inf / NaN / exponent-255 samples in some channels: the cascades that replay their block after meeting such a value must
    replay it from the INPUT, which the first pass has by then overwritten with outputs -- the library copies the input of an
    in-place call aside first.  Against the oracle, bit for bit, outputs and state."""
    import torch
    C = 21
    prog = pb.synth_program(fmt, C, sections, 0, gain=1.0)
    blocks = [128, 300, 64]
    n = sum(blocks)
    flt = fmt in (5, 6)
    x = pb.lcg_input(n, C, flt, seed=78)
    if flt:
        xi = x.view(np.uint32)
        xi[5, 2] = 0x7F800000; xi[127, 4] = 0xFFC00000; xi[128, 7] = 0x7F7FFFFF; xi[129, 7] = 0x7F7FFFFF; xi[400, 11] = 0xFFFFFFFF
        x[200:230, 15] = 3.0e38
    else:
        x[:, 3] = np.int32(0x7FFFFFFF); x[::2, 3] = np.int32(-0x7FFFFFFF)     # full scale through a cascade that amplifies
    o = po.OracleProgram(fmt, prog)
    r = rt.Runtime(fmt, prog)
    st = torch.cuda.current_stream().cuda_stream
    pos = 0
    for b in blocks:
        want = o.run_block(x[pos:pos + b], C, C)
        buf = dm.to_device(x[pos:pos + b].copy())
        r.run_block_device(buf.data_ptr(), C, C, buf.data_ptr(), C, 0, b, st)
        torch.cuda.synchronize()
        got = dm.to_host(buf)
        bad = np.nonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=0))[0]
        assert bad.size == 0, f"block at {pos}: channels {bad.tolist()} differ"
        pos += b
    assert (r.sync_state() == o.state).all()
    r.release()


def test_reset_keeps_store_mem_words_like_the_reference():
    """dspRuntimeReset zeroes the data area only (dsp_runtime.c:141): what DSP_STORE_MEM wrote into the
    program's parameter section is still there afterwards, also when the host never synced in between."""
    prog = np.fromfile(os.path.join(os.path.dirname(__file__), "golden", "dacdiy1.bin"), dtype=np.uint32)
    x = pb.lcg_input(64, 16, False, seed=9)
    o = po.OracleProgram(2, prog, fs=48000, random=3, dither=24)
    r = rt.Runtime(2, prog, fs=48000, random=3, dither=24)
    o.run_block(x, 32, 8, 0, scratch_len=40, block=64)
    r.run_block(x, 32, 8, 0, block=64)
    assert o.reset(96000, 5, 24) == 0 and r.reset(96000, 5, 24) == 0
    n = int(prog[1])
    assert (r.buf[:n] == o.buf[:n]).all() and (r.buf[:n] != prog[:n]).any()
    want = o.run_block(x, 32, 8, 0, scratch_len=40, block=64)
    got = r.run_block(x, 32, 8, 0, block=64)
    assert (got == want).all()
    assert (r.sync_state() == o.state).all()


@pytest.mark.parametrize("fmt", [2, 6])
def test_live_parameter_edits(fmt):
    """The reference reads parameters from the program words on every frame; a host may change a gain, a
    biquad coefficient, a tap or a bypass flag between two frames.  dspRuntimeUploadParams carries such
    edits to the device and keeps every filter's state (FIR histories included)."""
    taps = 0 if fmt == 2 else 40
    prog = pb.synth_program(fmt, 3, 3, taps)
    x = pb.lcg_input(300, 3, fmt == 6, seed=4)
    o = po.OracleProgram(fmt, prog)
    r = rt.Runtime(fmt, prog)
    assert_close(r.run_block(x[:100], 3, 3), o.run_block(x[:100], 3, 3), fmt)

    def find(op, nth=0):
        i, seen = 0, 0
        while True:
            word = int(prog[i]); code, skip = word >> 16, word & 0xFFFF
            if code == op:
                if seen == nth:
                    return i
                seen += 1
            i += skip

    lg = find(35, 1)                                   # DSP_LOAD_GAIN of channel 1: [io][offset -> gain word]
    gain_at = lg + int(np.int32(prog[lg + 2]))
    bq = find(50, 0)                                   # DSP_BIQUADS of channel 0: [state][offset -> bank]
    bank = bq + int(np.int32(prog[bq + 2]))
    edits = {gain_at: np.float32(0.37).view(np.uint32) if fmt == 6 else np.uint32(int(0.37 * (1 << 28))),
             bank + 5: prog[bank + 5] ^ np.uint32(0x00010000),          # b0 of the first section, slightly different
             find(50, 2) + int(np.int32(prog[find(50, 2) + 2])) + 1: np.uint32(0)}   # bypass the bank of channel 2
    if taps:
        fir = find(51, 1)
        imp = fir + int(np.int32(prog[fir + 1]))
        edits[imp + 3] = np.float32(0.125).view(np.uint32)               # one tap of channel 1
    for at, v in edits.items():
        r.buf[at] = v
        o.buf[at] = v
    r.upload_params()
    assert_close(r.run_block(x[100:200], 3, 3), o.run_block(x[100:200], 3, 3), fmt, what="after the edits")
    assert_close(r.run_block(x[200:], 3, 3), o.run_block(x[200:], 3, 3), fmt, what="next block")
    assert (r.sync_state() == o.state).all()


def test_live_parameter_edit_in_an_interpreted_core():
    prog = np.fromfile(os.path.join(os.path.dirname(__file__), "golden", "crossoverLV6.bin"), dtype=np.uint32)
    x = pb.lcg_input(120, 16, False, seed=12)
    o = po.OracleProgram(2, prog, fs=48000, random=2, dither=24)
    r = rt.Runtime(2, prog, fs=48000, random=2, dither=24)
    assert (r.run_block(x[:60], 32, 8, 0, block=30) == o.run_block(x[:60], 32, 8, 0, scratch_len=40, block=30)).all()
    i = 0
    while (int(prog[i]) >> 16) != 41:                  # first DSP_GAIN: [offset -> Q28 gain]
        i += int(prog[i]) & 0xFFFF
    at = i + int(np.int32(prog[i + 1]))
    r.buf[at] = o.buf[at] = np.uint32(1 << 26)         # 0.25
    r.upload_params()
    assert (r.run_block(x[60:], 32, 8, 0, block=30) == o.run_block(x[60:], 32, 8, 0, scratch_len=40, block=30)).all()
    assert (r.sync_state() == o.state).all()


def test_tagoutput_like_the_plugin():
    """linux/avdsp_plugin.c:133-137: the first output channel of each core carries (previoussample & 0xFF00) in bits 8..15;
    previoussample runs through cores (outer) and frames (inner) of a block and on into the next block."""
    prog = np.fromfile(os.path.join(os.path.dirname(__file__), "golden", "crossoverLV6.bin"), dtype=np.uint32)
    x = pb.lcg_input(600, 16, False, seed=21)
    o = po.OracleProgram(2, prog, fs=48000, random=3, dither=24)
    r = rt.Runtime(2, prog, fs=48000, random=3, dither=24)
    first_out = [25 - 0, 28 - 0]                    # cores' first output IO: usedOutputs 0x0E000000 and 0x30000000
    prev = 0
    assert rt.lib().dspRuntimeTagOutputReset(0) == -1           # nothing on the device yet
    for b0, b1 in ((0, 257), (257, 258), (258, 600)):
        want = np.zeros((b1 - b0, 32), dtype=np.int32)
        got = np.zeros((b1 - b0, 32), dtype=np.int32)
        for k, core in enumerate(o.cores):
            o.L.oracle_run_block(o.ctx, core, o.data_ptr, x[b0:b1].ctypes.data, 16, 8, want.ctypes.data, 32, 0, b1 - b0, 40)
            for n in range(b1 - b0):                             # the plugin's lines, frame by frame
                new = int(want[n, first_out[k]]) & -65536
                want[n, first_out[k]] = np.int32(new | (prev & 0xFF00))
                prev = ((new >> 8) + 0x100) & 0xFFFFFFFF
                prev = prev - (1 << 32) if prev & 0x80000000 else prev
            f = getattr(r.L, "dspRuntimeBlock_2")
            assert f(r.cores[k], r.rundata, x[b0:b1].ctypes.data, 16, 8, got.ctypes.data, 32, 0, b1 - b0) == 0
            r.tag_output(got, first_out[k])
        assert (got == want).all(), (b0, b1)


def test_kernel_timers_sample_every_nth_launch():
    """profile_stride n: only every n-th launch of a kind carries an event pair (bench.py's timed region uses 4)"""
    fmt, C = 6, 8
    prog = pb.synth_program(fmt, C, 2, 64)
    x = pb.lcg_input(512, C, True, seed=1)
    r = rt.Runtime(fmt, prog)
    r.set_option("profile", 1)
    r.set_option("profile_stride", 4)
    for _ in range(8):
        r.run_block(x, C, C)
    ms, n = r.kernel_time(1)
    assert n == 2 and ms > 0                                  # 8 FIR launches, every fourth bracketed
    r.set_option("profile_stride", 1)
    for _ in range(3):
        r.run_block(x, C, C)
    assert r.kernel_time(1)[1] == 3
    r.set_option("profile", 0)


# ---------------------------------------------------------------------------------------------
# pageable buffers of 1 MB and more go through the library's pinned chunks (avdsp_kernels.hip copy_from_caller / copy_to_caller)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("frames", [1021, 2051])
def test_pageable_blocks_of_odd_sizes_cross_in_chunks(frames):
    """blocks whose bytes are no multiple of the 4-MB chunk (and, 2051 frames, more than two chunks each way, several 1024-frame launches)
    through dspRuntimeBlock_N with ordinary numpy buffers -- and through buffers the CALLER has pinned (torch's pinned allocator), which the
    library recognises and copies without the detour: both the oracle's bits; then the state area (1.3 MB: the FIR histories) up and down"""
    torch = pytest.importorskip("torch")
    C, S, T = 1027, 2, 320
    prog = pb.synth_program(6, C, S, T)
    x = pb.lcg_input(frames, C, True, seed=frames)
    o = po.OracleProgram(6, prog)
    want = o.run_block(x, C, C)
    r = rt.Runtime(6, prog)
    got = r.run_block(x, C, C)
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    assert (r.sync_state() == o.state).all()
    r.release()
    xp = torch.empty(x.shape, dtype=torch.float32, pin_memory=True); xp.numpy()[:] = x
    yp = torch.zeros((frames, C), dtype=torch.float32, pin_memory=True)
    r = rt.Runtime(6, prog)
    got = r.run_block(xp.numpy(), C, C, out=yp.numpy())
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    # state up and down: what the host writes into the data area arrives, byte for byte, and comes back
    st = r.sync_state()
    assert st.nbytes > (1 << 20)
    rng = np.random.default_rng(5)
    pattern = rng.integers(0, 1 << 31, size=st.size, dtype=np.int64).astype(st.dtype)
    pattern &= 0x3FFFFFFF                                    # (finite floats, small ints: whatever a later block makes of them is not looked at)
    st[:] = pattern
    r.upload_state()
    st[:] = 0
    assert (r.sync_state() == pattern).all()
    r.release()
