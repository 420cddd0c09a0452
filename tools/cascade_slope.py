#!/usr/bin/env python3
"""Where does the cascade kernel's time go?  Launch time of the cascade kernel (biquad_row / biquad_row_i64 by default) against frames per block (the slope is the cost of a
step, the intercept what a launch costs besides stepping: loading coefficients and state, the pipeline's fill, the state
write-back) and against channels (waves per SIMD).   python tools/cascade_slope.py [fmt]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avdsp_amd import progbuilder as pb
from avdsp_amd import devmem as dm
from avdsp_amd import runtime as rt

fmt = int(sys.argv[1]) if len(sys.argv) > 1 else 6
S = 16
for C in (512, 4096, 8192):
    r = rt.Runtime(fmt, pb.synth_program(fmt, C, S))
    r.set_option("profile", 1)
    x = dm.to_device(pb.lcg_input(1024, C, fmt == 6))
    y = torch.zeros_like(x)
    st = torch.cuda.current_stream().cuda_stream
    rows = []
    for B in (16, 64, 128, 256, 512, 768, 1024):
        for _ in range(5):
            r.run_block_device(x.data_ptr(), C, C, y.data_ptr(), C, 0, B, st)
        torch.cuda.synchronize()
        r.kernel_time(0)
        for _ in range(20):
            r.run_block_device(x.data_ptr(), C, C, y.data_ptr(), C, 0, B, st)
        torch.cuda.synchronize()
        ms, n = r.kernel_time(0)
        rows.append((B, ms / n * 1e3))
    (b0, t0), (b1, t1) = rows[3], rows[-1]
    slope = (t1 - t0) / (b1 - b0)
    print(f"fmt {fmt}  C={C:5d} ({C * S // 64 / 1024:.2f} waves/SIMD): " + "  ".join(f"B={b}: {t:6.1f} us" for b, t in rows) +
          f"   slope {slope * 1e3:.1f} ns/step = {slope * 2.4e3:.0f} cycles @2.4GHz, intercept {t1 - slope * (b1 + 2 * S - 1):.1f} us")
    r.release()
