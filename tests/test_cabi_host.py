"""CPU-only checks of the C-ABI library: it loads, exports every symbol include/*.h declares, and the
host-side logic (validation, core lookup, lowering, refusals) behaves like the reference API says.
No compute call is made here: without a GPU the block entry points must FAIL (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from avdsp_amd import progbuilder as pb
from avdsp_amd import runtime as rt
from tests.golden_recipes import GOLDEN_DIR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _release():
    yield
    rt.lib().dspRuntimeRelease()


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", text)
    return [n for n in names if n.startswith(("dsp", "avdsp_hip_"))]


def test_library_exports_everything_the_headers_declare():
    L = rt.lib()
    fns = declared_functions("avdsp_runtime.h") + declared_functions("avdsp_hip.h")
    assert len(fns) > 30
    missing = [n for n in fns if not hasattr(L, n)]
    assert not missing, missing
    for name in rt.EXPORTED + ["dspHeaderPtr", "dspBiquadFreqSkip", "dspMantissa", "dspOpcodeText"]:
        assert hasattr(L, name), name


def test_library_exports_nothing_the_headers_do_not_declare():
    """The dynamic symbol table of the C ABI is the headers' functions + the reference's four data symbols and NOTHING else: no kernel
    stub (`show_through` and its `__device_stub__` sat there through round 4), no helper.  What hipcc itself adds to a code object's
    host side (`__hip_*`, `_fini/_init`, the `.hip_fatbin` handles) is the toolchain's, not the library's."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", rt.LIB_PATH], capture_output=True, text=True, check=True).stdout
    defined = {l.split()[-1] for l in out.splitlines() if len(l.split()) >= 3 and l.split()[-2] in "TDBRVWi"}
    ours = {n for n in defined if not n.startswith(("__hip", "_init", "_fini", "__bss_start", "_edata", "_end", "__odr_asan"))}
    declared = set(declared_functions("avdsp_runtime.h") + declared_functions("avdsp_hip.h"))
    declared |= {"dspHeaderPtr", "dspBiquadFreqSkip", "dspMantissa", "dspOpcodeText", "dspQNM", "dspQM64", "dspQM32"}
    extra = sorted(ours - declared)
    assert not extra, f"exported but not declared in include/*.h: {extra}"
    assert not [n for n in defined if "show_through" in n or "device_stub" in n]


def test_reference_data_symbols():
    L = rt.lib()
    txt = (C.c_char_p * 62).in_dll(L, "dspOpcodeText")
    assert txt[0] == b"DSP_END_OF_CODE" and txt[50] == b"DSP_BIQUADS" and txt[51] == b"DSP_FIR" and txt[61] == b"DSP_SINE"
    assert txt[3] == b"\nDSP_CORE"                         # dsp_header.c:10-73 keeps those newlines
    assert C.c_int.in_dll(L, "dspMantissa").value == 28
    # dspQNM family (dsp_header.h:276-285): truncation toward zero, saturation
    assert L.dspQM32(0.5, 28) == 1 << 27
    assert L.dspQM32(-0.75, 28) == -(3 << 26)
    assert L.dspQM32(9.0, 28) == 0x7FFFFFFF
    assert L.dspQM32(-9.0, 28) == -(1 << 31)
    assert L.dspQNM(1.0, 4, 28) == 1 << 28
    assert L.dspQM64(1.0, 40) == 1 << 40


def test_find_core_and_init_on_committed_program():
    prog = np.fromfile(os.path.join(GOLDEN_DIR, "crossoverLV6.bin"), dtype=np.uint32)
    r = rt.Runtime(2, prog, fs=48000, dither=24)
    assert r.rc == 288                                      # header.totalLength, as the reference returns
    assert len(r.cores) == 2
    base = r.buf.ctypes.data
    # CORE words at 121 and 156; execution starts behind them (core 2 also skips its PARAM)
    assert (r.cores[0] - base) // 4 == 124
    assert (r.cores[1] - base) // 4 == 269
    assert C.c_int.in_dll(r.L, "dspBiquadFreqSkip").value == 2 + 6 * 4      # 44.1k .. 96k encoded
    assert r.reset(192000) == -2 and r.reset(12345) == -1 and r.reset(96000) == 0


def test_lowering_reports_chains():
    r = rt.Runtime(6, pb.synth_program(6, 5, 3, 7))
    assert r.core_info() == dict(chains=5, max_sections=3, max_taps=7)
    r = rt.Runtime(2, pb.synth_program(2, 64, 16))
    assert r.core_info() == dict(chains=64, max_sections=16, max_taps=0)


def test_path_selection_and_refusals():
    """Host-only lowering: chains -> parallel kernels, anything else -> general interpreter (chains == 0),
    and a loud refusal where neither has a defined result."""
    prog = np.fromfile(os.path.join(GOLDEN_DIR, "crossoverLV6.bin"), dtype=np.uint32)
    r = rt.Runtime(2, prog, fs=48000, dither=24)
    assert r.core_info(0) == dict(chains=0, max_sections=0, max_taps=0)      # TPDF, X/Y ops, delay
    assert r.core_info(1)["chains"] == 0
    for name, fmt in (("tour_int.bin", 2), ("tour_float.bin", 3), ("tour_float.bin", 6)):
        r = rt.Runtime(fmt, np.fromfile(os.path.join(GOLDEN_DIR, name), dtype=np.uint32), fs=96000, dither=24)
        assert all(r.core_info(k)["chains"] == 0 for k in range(len(r.cores)))
    r = rt.Runtime(5, pb.synth_program(6, 4, 2, 7))          # formats 3 and 5: chains too (one lane per chain)
    assert r.core_info() == dict(chains=4, max_sections=2, max_taps=7)
    r = rt.Runtime(2, pb.synth_program(2, 2, 1, 9))          # FIR in int64 mode: undefined in the reference
    with pytest.raises(rt.AvdspError) as e:
        r.core_info()
    assert e.value.code == -8 and "undefined behaviour" in str(e.value)
    r = rt.Runtime(2, pb.synth_program(6, 2, 1))             # float-encoded program, int64 entry point: converted in place
    assert int(r.buf[6]) & 0xFFFF == 0                       # (dspChangeFormat, dsp_runtime.c:198-299; tests/test_changeformat.py)
    assert r.core_info()["chains"] == 2
    assert int(r.buf[6]) & 0xFFFF == 28
    # a chain that loads what another chain of the same core stores is a sequential dependency:
    # not parallel chains, so the frame-sequential interpreter takes it
    pw = pb.ProgramWriter(6)
    pw.core(); pw.load(4); pw.store(1); pw.load(1); pw.store(2)
    r = rt.Runtime(6, pw.end_of_code())
    assert r.core_info()["chains"] == 0
    # two chains storing the same IO: order matters, same answer
    pw = pb.ProgramWriter(6)
    pw.core(); pw.load(4); pw.store(1); pw.load(5); pw.store(1)
    r = rt.Runtime(6, pw.end_of_code())
    assert r.core_info()["chains"] == 0
    # offsets the opcode stream would follow outside the buffer are caught on the host
    prog = np.fromfile(os.path.join(GOLDEN_DIR, "dacdiy1.bin"), dtype=np.uint32).copy()
    i = 0
    while (int(prog[i]) >> 16) != 39:                       # DSP_LOAD_MEM: [program-relative offset]
        i += int(prog[i]) & 0xFFFF
    prog[i + 1] = 100000
    r = rt.Runtime(2, prog, fs=48000, dither=24)
    bad = []
    for k in range(len(r.cores)):
        try:
            r.core_info(k)
        except rt.AvdspError as e:
            bad.append((e.code, str(e)))
    assert len(bad) == 1 and bad[0][0] == -8 and "outside the program" in bad[0][1]


def test_shard_cut_is_host_side_and_works_on_any_program():
    """dspRuntimeSetShard / dspRuntimeShardInfo: contiguous balanced ranges of the lowered chain list, IO windows of the
    rank's own chains, on a program with scattered IO numbers; interpreter cores are not cut; bad arguments are refused."""
    L = rt.lib()
    pw = pb.ProgramWriter(6)
    pw.core()
    ios = [(40, 3), (17, 9), (33, 0), (18, 30), (50, 7)]
    for i, o in ios:
        pw.load(i); pw.sat0db(); pw.store(o)
    r = rt.Runtime(6, pw.end_of_code())
    assert r.shard_info() == dict(total_chains=5, first_chain=0, nchains=5, in_io_min=17, in_io_max=50, out_io_min=0, out_io_max=30)
    seen = []
    for rank in range(3):
        r.set_shard(rank, 3)
        info = r.shard_info()
        mine = ios[info["first_chain"]:info["first_chain"] + info["nchains"]]
        seen += mine
        assert info["in_io_min"] == min(i for i, _ in mine) and info["in_io_max"] == max(i for i, _ in mine)
        assert info["out_io_min"] == min(o for _, o in mine) and info["out_io_max"] == max(o for _, o in mine)
        assert r.get_option("shard_rank") == rank and r.get_option("shard_world") == 3
    assert seen == ios
    r.set_shard(6, 7)                                         # more ranks than chains: an empty slice, not an error
    assert r.shard_info()["nchains"] == 0 and r.shard_info()["total_chains"] == 5
    for bad in ((3, 3), (-1, 2), (0, 0)):
        assert L.dspRuntimeSetShard(*bad) == -1
    r.set_shard(0, 1)
    prog = np.fromfile(os.path.join(GOLDEN_DIR, "crossoverLV6.bin"), dtype=np.uint32)
    r = rt.Runtime(2, prog, fs=48000, dither=24)
    r.set_shard(1, 2)
    assert r.shard_info(0)["total_chains"] == 0               # not a chain core: runs whole on every rank
    r.set_shard(0, 1)


def test_no_cpu_fallback_without_a_gpu():
    if rt.lib().avdsp_hip_device_count() > 0:
        pytest.skip("a GPU is visible here")
    r = rt.Runtime(6, pb.synth_program(6, 2, 2))
    with pytest.raises(rt.AvdspError) as e:
        r.run_block(np.zeros((4, 2), dtype=np.float32), 2, 2)
    assert e.value.code == -10 and "HIP" in str(e.value)
    frame = np.zeros(4, dtype=np.float32)
    with pytest.raises(rt.AvdspError):
        r.run_frame(frame)


def test_programs_are_contexts_keyed_by_their_buffer():
    """host-only: two loaded programs, calls find theirs by pointer, the reference's exported data follow, release is per program"""
    import ctypes as C
    from avdsp_amd import progbuilder as pb
    from avdsp_amd import runtime as rt
    L = rt.lib()
    try:
        a = rt.Runtime(6, pb.synth_program(6, 8, 2, 40))
        b = rt.Runtime(2, pb.synth_program(2, 5, 3, 0))
        hdr = C.c_void_p.in_dll(L, "dspHeaderPtr")
        assert hdr.value == b.buf.ctypes.data
        ia = a.shard_info()                                   # names a core of program a: a becomes current
        assert hdr.value == a.buf.ctypes.data and ia["total_chains"] == 8
        assert b.shard_info()["total_chains"] == 5 and hdr.value == b.buf.ctypes.data
        a.set_shard(1, 2)
        assert a.shard_info()["nchains"] == 4 and a.shard_info()["first_chain"] == 4
        assert b.shard_info()["nchains"] == 5                 # b keeps its own (whole) shard
        assert L.dspRuntimeSelect(a.buf.ctypes.data + 16) == 0 and L.dspRuntimeGetOption(b"shard_rank") == 1
        assert L.dspRuntimeSelect(0) < 0
        a.release()
        assert L.dspRuntimeSelect(a.buf.ctypes.data) < 0 and L.dspRuntimeSelect(b.buf.ctypes.data) == 0
        assert L.dspRuntimeReleaseProgram(a.buf.ctypes.data) < 0
    finally:
        L.dspRuntimeSetShard(0, 1)
        L.dspRuntimeRelease()


def test_two_programs_back_to_back_in_one_array():
    """rows of one array, the second program starting where the first one's data area ends: both stay loaded and a
    pointer to the second program's first word names the second program (half-open word ranges)"""
    L = rt.lib()
    pa, pb_ = pb.with_data_area(pb.synth_program(2, 3, 2)), pb.with_data_area(pb.synth_program(6, 2, 1, 5))
    arr = np.concatenate([pa, pb_]).astype(np.uint32)
    a_ptr, b_ptr = arr.ctypes.data, arr.ctypes.data + 4 * len(pa)
    ra = L.dspRuntimeInit(a_ptr, len(pa), 48000, 0, 31)
    rb = L.dspRuntimeInit(b_ptr, len(pb_), 48000, 0, 31)
    assert ra == int(pa[1]) and rb == int(pb_[1])
    hdr = C.c_void_p.in_dll(L, "dspHeaderPtr")
    assert hdr.value == b_ptr
    assert L.dspRuntimeSelect(a_ptr) == 0 and C.c_void_p.in_dll(L, "dspHeaderPtr").value == a_ptr      # A survived B's load
    assert L.dspRuntimeSelect(b_ptr) == 0 and C.c_void_p.in_dll(L, "dspHeaderPtr").value == b_ptr      # ... and B's first word is B's
    assert L.dspRuntimeSelect(a_ptr + 4 * (len(pa) - 1)) == 0 and C.c_void_p.in_dll(L, "dspHeaderPtr").value == a_ptr
    assert L.dspFindCore(b_ptr, 1) and L.dspFindCore(a_ptr, 1)
    assert L.dspRuntimeReleaseProgram(a_ptr) == 0 and L.dspRuntimeReleaseProgram(b_ptr) == 0


def test_failed_loads_leave_no_context_behind():
    """a host probing buffers that hold no program must not fill the table of loaded programs (64)"""
    L = rt.lib()
    junk = [np.full(64, 0x12345678 + i, dtype=np.uint32) for i in range(70)]
    for j in junk:
        assert L.dspRuntimeInit(j.ctypes.data, len(j), 48000, 0, 31) == -1
    bad = pb.with_data_area(pb.synth_program(2, 2, 2))
    bad[3] ^= 1                                              # header.checkSum
    for _ in range(70):
        assert L.dspRuntimeInit(bad.ctypes.data, len(bad), 48000, 0, 31) == -4
    good = rt.Runtime(2, pb.synth_program(2, 2, 2))
    assert good.rc > 0
    # a failed load does not take the current program away from calls that name none
    assert L.dspRuntimeInit(junk[0].ctypes.data, 64, 48000, 0, 31) == -1
    assert C.c_void_p.in_dll(L, "dspHeaderPtr").value == good.buf.ctypes.data


def test_strand_info_reports_what_lowers():
    """host-only: dspRuntimeStrandInfo -- the run of identical strands a core ends in, the opcode words in front of it, and the
    floor of 65 strands ("strand_lanes" 2 lowers shorter runs too, 0 none)"""
    from tests.test_gpu_strands import crossover_program
    r = rt.Runtime(6, crossover_program(12, 6, "dither"), fs=48000, dither=24)
    try:
        assert r.strand_info(0)["strands"] == 0                 # up to 64 strands: the interpreter's strand groups are faster
        r.set_option("strand_lanes", 2)
        info = r.strand_info(0)
        assert info["strands"] == 12 and info["ops"] == 6 and info["prefix_words"] > 0      # the TPDF_CALC stays in front
        r.set_option("strand_lanes", 0)
        assert r.strand_info(0)["strands"] == 0
        r.set_option("strand_lanes", 1)
        r.release()
        r = rt.Runtime(6, crossover_program(100, 6, "dither"), fs=48000, dither=24)
        assert r.strand_info(0)["strands"] == 100
    finally:
        r.set_option("strand_lanes", 1)
        r.release()
    prog = np.fromfile(os.path.join(GOLDEN_DIR, "dacdiy1.bin"), dtype=np.uint32)
    r = rt.Runtime(2, prog, fs=48000, dither=24)
    try:
        assert [r.strand_info(i)["strands"] for i in range(len(r.cores))] == [0, 0, 0, 0]      # runs of two: the interpreter's
        r.set_option("strand_lanes", 2)
        assert [r.strand_info(i)["strands"] for i in range(len(r.cores))] == [2, 2, 2, 0]
    finally:
        r.set_option("strand_lanes", 1)
        r.release()
