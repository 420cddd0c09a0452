import os, sys
sys.path.insert(0, '.')
import numpy as np
from avdsp_amd import progbuilder as pb, runtime as rt
name, fmt, in_stride, in_base, out_stride = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
prog = np.fromfile("tests/golden/" + name, dtype=np.uint32)
x = pb.lcg_input(256, in_stride, fmt in (5, 6), seed=5)
r = rt.Runtime(fmt, prog, fs=48000, random=1, dither=24)
r.run_block_all(x, out_stride, in_base, 0, block=256)
