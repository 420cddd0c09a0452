#!/usr/bin/env python3
"""What dspRuntimeSetOption("fir_split", 1) costs in accuracy: cfg4's shape (256 ch x 4096 taps, blocks of 1024 frames, DSP_FORMAT 6),
five blocks, against the same blocks with the option off (the reference's bits, tests/test_gpu_headline.py): words that differ, the
largest difference relative to the block's peak and in units of the last place.   python tools/fir_split_error.py   (GPU box)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from avdsp_amd import progbuilder as pb      # noqa: E402
from avdsp_amd import runtime as rt          # noqa: E402

C, T, B, nb = 256, 4096, 1024, 5
prog = pb.synth_program(6, C, 0, T)
x = pb.lcg_input(B * nb, C, True, seed=20260104)
outs = {}
for split in (0, 1):
    r = rt.Runtime(6, prog)
    r.set_option("fir_split", split)
    outs[split] = np.concatenate([r.run_block(x[k * B:(k + 1) * B], C, C) for k in range(nb)])
    r.set_option("fir_split", 0)
    r.release()
a, b = outs[0].astype(np.float64), outs[1].astype(np.float64)
diff = np.abs(a - b)
peak = np.abs(a).max()
wa, wb = outs[0].view(np.int32).astype(np.int64), outs[1].view(np.int32).astype(np.int64)
ulps = np.abs(wa - wb)[np.sign(a) == np.sign(b)]
print(f"fir_split 1 vs 0 on {C} ch x {T} taps, {nb} blocks of {B}: {np.count_nonzero(outs[0].view(np.uint32) != outs[1].view(np.uint32))} of {a.size} output words differ; "
      f"largest difference {diff.max():.3e} = {diff.max() / peak:.3e} of the peak ({peak:.3f}); in units of the last place: max {ulps.max()}, "
      f"words one ulp apart {np.count_nonzero(ulps == 1)}, more {np.count_nonzero(ulps > 1)}")
