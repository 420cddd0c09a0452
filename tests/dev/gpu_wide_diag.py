"""Development probe: one wide-core random program through dspRuntimeBlockAll with and without strand groups."""
import sys; sys.path.insert(0, '.')
import numpy as np
from avdsp_amd import progbuilder as pb, runtime as rt
from oracle import pyoracle as po
from tests.fuzz_programs import IN_BASE, N_IN, N_OUT, random_program
seed, fmt, frames = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
prog = random_program(seed, fmt)
fs = [48000, 48000, 96000][seed % 3]
x = pb.lcg_input(frames, N_IN, fmt in (5, 6), seed=seed)
o = po.OracleProgram(fmt, prog, fs=fs, random=seed, dither=24)
want = o.run_block(x, N_OUT, IN_BASE, 0, block=frames, frame=np.zeros(4096, dtype=np.uint32))
for split in (0, 1):
    r = rt.Runtime(fmt, prog, fs=fs, random=seed, dither=24)
    r.set_option("strand_split", split)
    got = r.run_block_all(x, N_OUT, IN_BASE, 0, block=frames)
    cols = np.nonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=0))[0]
    rows = np.nonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=1))[0]
    print("split", split, "pieces", r.get_option("pieces"), "levels", r.get_option("levels"), "cols", list(cols), "first rows", list(rows[:4]))
    if cols.size:
        c = int(cols[0]); f = int(rows[0])
        print("   got ", [hex(int(v)) for v in got.view(np.uint32)[f:f + 4, c]], "\n   want", [hex(int(v)) for v in want.view(np.uint32)[f:f + 4, c]])
    r.set_option("strand_split", 1); r.release()
