"""Development probe: one random program (seed, fmt) through the interpreter variants against the oracle.
usage: python tests/dev/gpu_seed_diag.py SEED FMT [block]"""
import sys; sys.path.insert(0, '.')
import numpy as np
from avdsp_amd import progbuilder as pb, runtime as rt
from oracle import pyoracle as po
from tests.fuzz_programs import IN_BASE, N_IN, N_OUT, random_program
seed, fmt = int(sys.argv[1]), int(sys.argv[2])
block = int(sys.argv[3]) if len(sys.argv) > 3 else 2
frames = 300
prog = random_program(seed, fmt)
fs = [48000, 48000, 96000][seed % 3]
x = pb.lcg_input(frames, N_IN, fmt in (5, 6), seed=seed)
for mode in ("all", "percore"):
    for impl in (1, 0):
        o = po.OracleProgram(fmt, prog, fs=fs, random=seed, dither=24)
        want = o.run_block(x, N_OUT, IN_BASE, 0, block=block, frame=np.zeros(4096, dtype=np.uint32))
        r = rt.Runtime(fmt, prog, fs=fs, random=seed, dither=24)
        r.set_option("interp_impl", impl)
        got = (r.run_block_all if mode == "all" else r.run_block)(x, N_OUT, IN_BASE, 0, block=block)
        r.sync_state(); nn = int(prog[1]) + int(prog[2])
        cols = np.nonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=0))[0]
        rows = np.nonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=1))[0]
        words = np.nonzero(r.buf[12:nn] != o.buf[12:nn])[0] + 12
        print(mode, "impl", impl, "cols", list(cols), "first frame", rows[:1], "words", list(words[:6]))
        if cols.size:
            f = int(rows[0]); c = int(cols[0])
            print("   got", got.view(np.uint32)[f:f + 3, c], "want", want.view(np.uint32)[f:f + 3, c])
            print("   state got", [hex(int(v)) for v in r.buf[words[:6]]], "want", [hex(int(v)) for v in o.buf[words[:6]]])
        r.set_option("interp_impl", 1); r.release()
