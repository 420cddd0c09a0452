"""avdsp_amd.progbuilder against programs emitted by the reference encoder library (fixtures made by
tests/golden/make_goldens.py through oracle/ref_encode.c), and the loader-level format checks on the
.bin files the reference itself commits (osx/*.bin copied to tests/golden/ as data)."""
import glob
import hashlib
import os

import numpy as np
import pytest

from avdsp_amd import progbuilder as pb
from tests.golden_recipes import GOLDEN_DIR


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_byte_identical_to_reference_encoder(manifest):
    for e in manifest["reference_encoder"]:
        w = pb.synth_program(e["fmt"], e["channels"], e["sections"], 0, e["fmin"], e["fmax"])
        assert len(w) == e["words"]
        assert sha(w) == e["sha"], e
    for path in glob.glob(os.path.join(GOLDEN_DIR, "refenc_*.npy")):
        ref = np.load(path)
        _, f, c, s, fmin, fmax = os.path.basename(path)[:-4].split("_")
        w = pb.synth_program(int(f[1:]), int(c[1:]), int(s[1:]), 0, int(fmin), int(fmax))
        assert (w == ref).all(), path


def test_dspcreate_reproduces_committed_bins(manifest):
    assert all(r["identical"] for r in manifest["dspcreate_reproduces"])


@pytest.mark.parametrize("name", ["crossoverLV6.bin", "dacdiy1.bin", "dsptest1.bin", "dacfabriceo.bin", "mydspcode.bin"])
def test_committed_programs_pass_checksum_walk(name):
    w = np.fromfile(os.path.join(GOLDEN_DIR, name), dtype=np.uint32)
    assert (int(w[0]) >> 16) == pb.OP_HEADER
    total, cores = pb.checksum(w)
    assert total == int(w[3])
    assert cores == int(w[4])
    assert int(w[1]) == len(w)


def test_fir_program_layout():
    """FIR opcode must address the LENGTH word (dsp_runtime.c:935-939), state must cover `length` words."""
    w = pb.synth_program(6, 2, 1, 7)
    pos = 0
    firs = []
    while True:
        skip, op = int(w[pos]) & 0xFFFF, int(w[pos]) >> 16
        if skip == 0:
            break
        if op == pb.OP_FIR:
            firs.append(pos)
        pos += skip
    assert len(firs) == 2
    for p in firs:
        off = int(np.int32(w[p + 1]))
        assert int(w[p + off]) == 7                    # length word, hi16 == 0
        assert (p + off) & 1                           # odd index, taps 8-byte aligned
    assert int(w[2]) == 27                             # 6+7, pad to even, 6+7 (addDataSpaceAligned8, dsp_encoder.c:141-144)
