"""The program encoder (avdsp_amd/csrc/avdsp_encoder.c) against bytes written by the
reference encoder.  CPU only.

  * oracle/enc_sweep.c and oracle/ref_encode_ops.c are programs written against the encoder API; the
    fixtures tests/golden/enc_sweep_*.bin and tour_*.bin are what the REFERENCE encoder made of them
    (tests/golden/make_goldens.py).  Here the same sources are linked with this repository's encoder
    library and must produce identical files.
  * where /root/reference exists (the build container), the reference's own DSP programs are compiled
    against this encoder and must reproduce the committed osx/*.bin byte for byte.
  * the FIR layout (where the reference encoder is wrong, see include/avdsp_encoder.h) is held to
    avdsp_amd/progbuilder.py, whose FIR programs the reference RUNTIME executes in the golden cases."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from avdsp_amd import encoder as enc
from avdsp_amd import progbuilder as pb
from tests.golden_recipes import GOLDEN_DIR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = [f"-I{ROOT}/include/compat", f"-I{ROOT}/include"]
LINK = [f"-L{ROOT}/avdsp_amd/lib", "-lavdsp_encoder", f"-Wl,-rpath,{ROOT}/avdsp_amd/lib", "-lm"]
REFERENCE = os.environ.get("AVDSP_REFERENCE", "/root/reference") + "/module_avdsp"


def build_exe(tmp_path, name, sources):
    exe = str(tmp_path / name)
    subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-w", *INC, "-o", exe, *sources, *LINK])
    return exe


def same_file(a, b):
    x, y = np.fromfile(a, dtype=np.uint32), np.fromfile(b, dtype=np.uint32)
    assert len(x) == len(y), f"{len(x)} words, reference encoder wrote {len(y)}"
    diff = np.nonzero(x != y)[0]
    assert diff.size == 0, f"{diff.size} words differ from the reference encoder's, first at {diff[:8]}"


def test_library_exports_the_declared_api():
    text = open(os.path.join(ROOT, "include", "avdsp_encoder.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(dsp[A-Za-z0-9_]*|addCode|addFloat|opcodeIndex[A-Za-z0-9]*|setSerialHash)\s*\(", text))
    names -= {"dspprintf", "dspprintf1", "dspprintf2", "dspprintf3"}
    assert len(names) > 150
    L = enc.lib()
    missing = sorted(n for n in names if not hasattr(L, n))
    assert not missing, missing


@pytest.mark.parametrize("variant", [(2, 4, 9), (6, 4, 7), (6, 0, 13), (2, 5, 5)], ids=lambda v: "fmt%d_f%d_%d" % v)
def test_encoder_sweep_identical_to_reference_encoder(tmp_path, variant):
    """All filter types and presets, tables, delays, meters ... : 3.4k-16k words per variant."""
    exe = build_exe(tmp_path, "enc_sweep", [f"{ROOT}/oracle/enc_sweep.c"])
    out = str(tmp_path / "sweep.bin")
    subprocess.check_call([exe, *map(str, variant), out], stdout=subprocess.DEVNULL)
    same_file(out, os.path.join(GOLDEN_DIR, "enc_sweep_%d_%d_%d.bin" % variant))


@pytest.mark.parametrize("variant", [(2, 4, 9), (6, 4, 7), (6, 0, 13), (2, 5, 5)], ids=lambda v: "fmt%d_f%d_%d" % v)
def test_hilbert_banks_identical_to_reference_encoder(tmp_path, variant):
    """dsp_Hilbert (encoder/dsp_filters.h:76; design encoder/dsp_HilbertDesign.c): both branches for 1 .. 10 stages at ten transition
    widths, every encoded rate -- the same bytes as the reference encoder (tests/golden/make_hilbert_goldens.py), Q28 and float."""
    exe = build_exe(tmp_path, "enc_hilbert", [f"{ROOT}/oracle/enc_hilbert.c"])
    out = str(tmp_path / "hilbert.bin")
    subprocess.check_call([exe, *map(str, variant), out], stdout=subprocess.DEVNULL)
    same_file(out, os.path.join(GOLDEN_DIR, "enc_hilbert_%d_%d_%d.bin" % variant))


@pytest.mark.parametrize("stages,width,fs_index,ripple", [(4, 160.0, pb.F48000, 0.5), (3, 120.0, pb.F44100, 1.5), (8, 40.0, pb.F96000, 0.05), (10, 500.0, pb.F48000, 1e-3)])
def test_hilbert_pair_is_a_90_degree_splitter(stages, width, fs_index, ripple):
    """What the coefficients are FOR, independent of any fixture: the reference branch behind a one-sample delay (as the reference's
    own program uses it, dspprogs/oktodac_fabriceo.c:204-208) and the +90 branch differ in phase by 90 degrees over
    [width, fs/2 - width], both all-pass.  Cells are (c - z^-2) / (1 - c z^-2)."""
    fs = {pb.F44100: 44100.0, pb.F48000: 48000.0, pb.F96000: 96000.0}[fs_index]

    def bank(phase):
        def build(L):
            L.dsp_PARAM()
            L.dspBiquad_Sections(stages)
            L.dsp_Hilbert(stages, C.c_double(width), C.c_float(phase))
            L.dsp_CORE()
            L.dsp_LOAD(1); L.dsp_STORE(0)
        words = enc.encode(build, 6, fs_index, fs_index)
        f = words.view(np.float32)
        # the bank: [BIQUADS head][per cell: FHILB head, Q, gain, pad, b0 b1 b2 a1-1 a2 (+pad)]: find the cells by their -1.0 / c pattern
        cs = [float(f[i]) for i in range(len(f) - 4) if f[i + 2] == -1.0 and f[i + 1] == 0.0 and f[i + 3] == -1.0 and f[i] == f[i + 4] and 0.0 < f[i] < 1.0]
        assert len(cs) == stages, cs
        return cs

    ref, quad = bank(0.0), bank(90.0)
    assert all(a < b for a, b in zip(ref, ref[1:])) and all(a < b for a, b in zip(quad, quad[1:]))
    freqs = np.linspace(width, fs / 2 - width, 400)
    z = np.exp(-1j * 2 * np.pi * freqs / fs)                 # z^-1
    H = lambda cs: np.prod([(c - z * z) / (1 - c * z * z) for c in cs], axis=0)
    hr, hq = H(ref) * z, H(quad)
    assert np.allclose(np.abs(hr), 1.0, atol=1e-6) and np.allclose(np.abs(hq), 1.0, atol=1e-6)
    dphi = np.degrees(np.angle(hq / hr))
    # equiripple: the bound is the design's (order and transition width), e.g. +-1.24 degrees for (3 stages, 120 Hz at 44.1 kHz)
    assert np.all(np.abs(np.abs(dphi) - 90.0) < ripple), (dphi.min(), dphi.max())


@pytest.mark.parametrize("fmt,name", [(2, "tour_int.bin"), (6, "tour_float.bin")])
def test_opcode_tour_identical_to_reference_encoder(tmp_path, fmt, name):
    exe = build_exe(tmp_path, "tour", [f"{ROOT}/oracle/ref_encode_ops.c"])
    out = str(tmp_path / name)
    subprocess.check_call([exe, str(fmt), out], stdout=subprocess.DEVNULL)
    same_file(out, os.path.join(GOLDEN_DIR, name))


@pytest.mark.parametrize("fmt,c,s,fmin,fmax", [(2, 8, 8, 5, 5), (6, 8, 8, 5, 5), (4, 3, 5, 4, 7), (2, 100, 16, 4, 9)])
def test_cascade_program_identical_to_progbuilder(tmp_path, fmt, c, s, fmin, fmax):
    """oracle/ref_encode.c through this encoder == progbuilder.py, which test_progbuilder.py holds
    byte-identical to the reference encoder's output for the same call."""
    exe = build_exe(tmp_path, "cascade", [f"{ROOT}/oracle/ref_encode.c"])
    out = str(tmp_path / "c.bin")
    subprocess.check_call([exe, str(fmt), str(c), str(s), str(fmin), str(fmax), out], stdout=subprocess.DEVNULL)
    got = np.fromfile(out, dtype=np.uint32)
    want = pb.synth_program(fmt, c, s, 0, fmin, fmax)
    assert len(got) == len(want) and (got == want).all()


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference sources only exist in the build container")
@pytest.mark.parametrize("prog,binname,args", [
    ("crossoverLV6", "crossoverLV6.bin", "-dspformat 2 -fsmax 96000 -fx 800"),
    ("oktodac_diy", "dacdiy1.bin", "-dspformat 2 -fsmax 192000 -prog 1 -dither 24"),
    ("testfunction", "dsptest1.bin", "-dspformat 3 -fsmax 96000 -test1 -dither 26"),
])
def test_reference_programs_reproduce_committed_bins(tmp_path, prog, binname, args):
    """osx/oktodac.mak:35-41: the reference's DSP programs, compiled from where they lie against THIS
    encoder, give the committed osx/*.bin (copied as fixtures) byte for byte."""
    exe = build_exe(tmp_path, prog, [f"{ROOT}/tests/csrc/enc_driver.c", f"{REFERENCE}/dspprogs/{prog}.c"])
    out = str(tmp_path / binname)
    subprocess.check_call([exe, out, *args.split()], stdout=subprocess.DEVNULL)
    same_file(out, os.path.join(GOLDEN_DIR, binname))


@pytest.mark.parametrize("fmt,channels,sections,taps,fmin,fmax", [(6, 3, 2, 7, 5, 5), (4, 2, 0, 33, 4, 6), (6, 2, 3, 8, 5, 7)])
def test_fir_programs_match_progbuilder(fmt, channels, sections, taps, fmin, fmax):
    """dsp_FIR with the impulse pointer on the LENGTH word: the layout the reference runtime reads
    (dsp_runtime.c:928-969) and progbuilder.py emits; the reference encoder's own dsp_FIR is off by one."""
    nf = fmax - fmin + 1
    all_taps = pb.lcg_taps_all(channels, taps)

    def build(L):
        L.dsp_CORE()
        for c in range(channels):
            L.dsp_PARAM()
            bank = 0
            if sections:
                bank = L.dspBiquad_Sections(sections)
                for sct in pb.synth_sections(c, sections, fmin, fmax):
                    L.dsp_Filter2ndOrder(sct.ftype, float(sct.freq), float(sct.q), float(sct.gain))
            imp = L.dspFir_Impulses()
            t = np.ascontiguousarray(all_taps[c], dtype=np.float32)
            for _ in range(nf):
                L.dspFir_ImpulseData(t.ctypes.data_as(C.POINTER(C.c_float)), taps)
            L.dsp_LOAD_GAIN_Fixed(channels + c, 1.0)
            if sections:
                L.dsp_BIQUADS(bank)
            L.dsp_FIR(imp)
            L.dsp_SAT0DB()
            L.dsp_STORE(c)

    got = enc.encode(build, fmt, fmin, fmax)
    want = pb.synth_program(fmt, channels, sections, taps, fmin, fmax)
    assert len(got) == len(want)
    assert (got == want).all(), np.nonzero(got != want)[0][:8]


def test_malformed_program_is_fatal_like_the_reference():
    """dsp_encoder.c:58-61: a malformed program prints FATAL ERROR and exits with status 1."""
    code = ("import numpy as np; from avdsp_amd import encoder as e; L = e.lib(); t = np.zeros(1000, dtype=np.uint32);"
            "L.dspEncoderInit(t.ctypes.data, 1000, 2, 5, 5, 8); L.dsp_CORE(); L.dsp_STORE(8)")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 1 and "FATAL ERROR : IO out of range." in r.stderr
    code = code.replace("L.dsp_STORE(8)", "L.dsp_BIQUADS(40)")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 1 and "FATAL ERROR" in r.stderr
