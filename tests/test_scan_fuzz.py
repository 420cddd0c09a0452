"""Memory safety of what the host accepts (no GPU needed): programs with damaged words -- offsets, counts, IO
numbers, skips, STORE_MEM targets; the checksum only covers head words, and the harness re-computes it anyway --
go through dspRuntimeInit + dspRuntimeCoreInfo (the host's chain lowering and opcode scan, avdsp_host.c).
Whatever the scan ACCEPTS is then run in the oracle built with AddressSanitizer, on exact-size heap buffers:
the device interpreter follows the same offsets, so a stray access there would be a GPU fault here.  The
reference trusts its encoder (a damaged program makes it read and write anywhere); a device kernel must not.
1400+ mutants per seed while developing; a bounded sample runs here."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from avdsp_amd import progbuilder as pb
from tests.fuzz_programs import random_program
from tests.golden_recipes import GOLDEN_DIR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = str(tmp_path_factory.mktemp("scanfuzz") / "scan_fuzz_driver")
    lib = os.path.join(ROOT, "avdsp_amd", "lib")
    cmd = ["gcc", "-fsanitize=address", "-g", "-O1", "-std=gnu99", "-I" + os.path.join(ROOT, "include"), "-o", exe,
           os.path.join(ROOT, "tests", "csrc", "scan_fuzz_driver.c"), os.path.join(ROOT, "oracle", "avdsp_oracle.c"),
           "-L" + lib, "-lavdsp_mi355x", "-Wl,-rpath," + lib, "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("AddressSanitizer build not available here: " + r.stderr[-300:])
    return exe


def sources():
    out = []
    for name, fmt in (("crossoverLV6.bin", 2), ("dacdiy1.bin", 2), ("tour_float.bin", 6), ("tour_int.bin", 2),
                      ("tour_float.bin", 3), ("dacfabriceo.bin", 2), ("mydspcode.bin", 2)):
        out.append((np.fromfile(os.path.join(GOLDEN_DIR, name), dtype=np.uint32), fmt, name))
    for seed in range(4):
        for fmt in (2, 6, 5):
            out.append((random_program(seed, fmt), fmt, f"random{seed}"))
    out.append((pb.synth_program(6, 3, 2, 9), 6, "chains"))          # chain lowering (lower_core)
    out.append((pb.synth_program(2, 4, 3), 2, "chains_int"))
    return out


def mutate(rng, prog):
    p = prog.copy()
    total = int(p[1])
    for _ in range(int(rng.integers(1, 4))):
        i = int(rng.integers(12, total))
        kind = int(rng.integers(0, 5))
        if kind == 0: v = int(rng.integers(-4, 70000))
        elif kind == 1: v = int(rng.integers(0, 2**32))
        elif kind == 2: v = int(p[i]) ^ (1 << int(rng.integers(0, 32)))
        elif kind == 3: v = int(p[i]) + int(rng.integers(-300, 300))
        else: v = int(rng.choice([0, 1, -1, 0x7FFFFFFF, 0x80000000, 65535, 65536, total, int(p[2]), int(p[2]) - 1]))
        p[i] = np.uint32(v & 0xFFFFFFFF)
    s, cores = pb.checksum(p[:total])
    p[3] = s
    p[4] = cores
    return p


@pytest.mark.parametrize("seed", [1, 2])
def test_accepted_programs_stay_inside_their_buffers(driver, tmp_path, seed):
    rng = np.random.default_rng(seed)
    srcs = sources()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    path = str(tmp_path / "mutant.bin")
    ran = refused = 0
    for it in range(120):
        prog, fmt, name = srcs[int(rng.integers(len(srcs)))]
        p = mutate(rng, prog)
        p.tofile(path)
        r = subprocess.run([driver, path, str(fmt), "48000", "70"], capture_output=True, text=True, env=env, timeout=120)
        assert r.returncode == 0, f"mutant {it} of {name} (DSP_FORMAT {fmt}): {r.stderr[-1500:]}"
        if r.stdout.startswith("ran"):
            ran += 1
        else:
            refused += 1
    assert ran > 20 and refused > 20            # the sample exercises both outcomes


def test_unmutated_programs_are_accepted(driver, tmp_path):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    for prog, fmt, name in sources():
        path = str(tmp_path / "plain.bin")
        prog.tofile(path)
        r = subprocess.run([driver, path, str(fmt), "48000", "70"], capture_output=True, text=True, env=env, timeout=120)
        assert r.returncode == 0 and r.stdout.startswith("ran"), (name, fmt, r.stdout, r.stderr[-500:])
