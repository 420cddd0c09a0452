"""Development sweep: random programs on the general interpreter, frame-parallel kernel where the host
allows it, against the oracle; reports how many block calls each kernel took.
usage: python tests/dev/gpu_wave_sweep.py SEED0 SEED1 [frames] [all]     (all: through dspRuntimeBlockAll)"""
import sys; sys.path.insert(0, '.')
import numpy as np
from avdsp_amd import progbuilder as pb, runtime as rt
from oracle import pyoracle as po
from tests.fuzz_programs import IN_BASE, N_IN, N_OUT, random_program

frames = int(sys.argv[3]) if len(sys.argv) > 3 else 300
use_all = len(sys.argv) > 4 and sys.argv[4] == "all"
import os
BLOCKS = [int(b) for b in os.environ.get("AVDSP_SWEEP_BLOCKS", "1,64,0").split(",")]      # 0 = all frames in one block
# AVDSP_SWEEP_OVERLAP=1: an output window IO 0 .. 47 that contains the input window (IO 32 .. 39): the shared columns show the input
# unless the program stores them (show_through in front of the call's launches)
OUT_STRIDE = 48 if os.environ.get("AVDSP_SWEEP_OVERLAP") == "1" else N_OUT
levels_hist = {}
bad = n = wave = scalar = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    for fmt in (2, 3, 4, 5, 6):
        prog = random_program(seed, fmt)
        fs, block = [48000, 48000, 96000][seed % 3], BLOCKS[seed % 3] or frames
        x = pb.lcg_input(frames, N_IN, fmt in (5, 6), seed=seed)
        o = po.OracleProgram(fmt, prog, fs=fs, random=seed, dither=24)
        r = rt.Runtime(fmt, prog, fs=fs, random=seed, dither=24)
        if r.rc < 0:
            continue
        r.set_option("profile", 1)
        if use_all:
            block = max(block, 2)
            want = o.run_block(x, OUT_STRIDE, IN_BASE, 0, block=block, frame=np.zeros(4096, dtype=np.uint32))
            got = r.run_block_all(x, OUT_STRIDE, IN_BASE, 0, block=block)
            key = (r.get_option("cores"), r.get_option("levels")); levels_hist[key] = levels_hist.get(key, 0) + 1
        else:
            want = o.run_block(x, OUT_STRIDE, IN_BASE, 0, scratch_len=48, block=block)
            got = r.run_block(x, OUT_STRIDE, IN_BASE, 0, block=block)
        r.sync_state(); nn = int(prog[1]) + int(prog[2]); n += 1
        w = r.kernel_time(5)[1]; s = r.kernel_time(3)[1]
        wave += w; scalar += s
        cols = np.nonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=0))[0]
        words = np.nonzero(r.buf[12:nn] != o.buf[12:nn])[0] + 12
        if cols.size or words.size:
            bad += 1
            first = int(np.nonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=1))[0][0]) if cols.size else -1
            print('MISMATCH seed', seed, 'fmt', fmt, 'block', block, 'wave/scalar launches', w, s, 'cols', list(cols), 'first frame', first,
                  'words', list(words[:8]), flush=True)
        r.set_option("profile", 0)
        r.release()
print('runs', n, 'bad', bad, 'frame-parallel launches', wave, 'scalar launches', scalar)
if use_all:
    print('(cores, levels) histogram:', sorted(levels_hist.items()))
