"""Development probe: state words of one strand, frame by frame, oracle vs device (NaN-heavy seed 10021 fmt 3)."""
import sys; sys.path.insert(0, '.')
import numpy as np
from avdsp_amd import progbuilder as pb, runtime as rt
from oracle import pyoracle as po
from tests.fuzz_programs import IN_BASE, N_IN, N_OUT, random_program
seed, fmt = 10021, int(sys.argv[1]) if len(sys.argv) > 1 else 3
prog = random_program(seed, fmt)
total = int(prog[1])
fs = [48000, 48000, 96000][seed % 3]
x = pb.lcg_input(8, N_IN, fmt in (5, 6), seed=seed)
o = po.OracleProgram(fmt, prog, fs=fs, random=seed, dither=24)
r = rt.Runtime(fmt, prog, fs=fs, random=seed, dither=24)
r.set_option("interp_impl", 0)
frame = np.zeros(4096, dtype=np.uint32)
words = [total + k for k in (130, 132, 133, 134, 138, 139, 140)]
for f in range(5):
    want = o.run_block(x[f:f + 1], N_OUT, IN_BASE, 0, block=1, frame=frame)
    got = r.run_block(x[f:f + 1], N_OUT, IN_BASE, 0, block=1)
    r.sync_state()
    print("frame", f, "out8", hex(int(got.view(np.uint32)[0, 8])), hex(int(want.view(np.uint32)[0, 8])))
    print("   dev   ", [hex(int(r.buf[w])) for w in words])
    print("   oracle", [hex(int(o.buf[w])) for w in words])
