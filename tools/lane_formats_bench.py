#!/usr/bin/env python3
"""DSP_FORMAT 3 and 5 (float accumulator, truncating dspMulFloatFloat) on the chain shapes of BASELINE's configs: the cascade
case (4096 ch x 16 sections) and the FIR case (256 ch x 4096 taps), blocks of 1024 frames resident in HBM; kernel time from the
library's timers.  Run on the GPU box:  python tools/lane_formats_bench.py [--blocks N]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from avdsp_amd import progbuilder as pb      # noqa: E402
from avdsp_amd import devmem as dm
from avdsp_amd import runtime as rt          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--blocks", type=int, default=3)
ap.add_argument("--frames", type=int, default=1024)
ap.add_argument("--cases", nargs="*", default=["bq", "fir", "north"])
ap.add_argument("--lane-hw", type=int, default=1, help="1: v_mul_f32 under round-toward-zero where it is the reference's product (round 4); 0: the integer products throughout")
args = ap.parse_args()
CASES = {"bq": (4096, 16, 0), "fir": (256, 0, 4096), "north": (512, 16, 4096)}
B = args.frames
for fmt in (3, 5):
    for case in args.cases:
        C, S, T = CASES[case]
        r = rt.Runtime(fmt, pb.synth_program(fmt, C, S, T))
        r.set_option("lane_hw", args.lane_hw)
        info = r.shard_info()
        x = dm.to_device(np.ascontiguousarray(pb.lcg_input(B, C, fmt == 5)))
        y = torch.zeros((B, C), dtype=x.dtype, device="cuda")
        run = lambda: r.run_block_device(x.data_ptr(), C, info["in_io_min"], y.data_ptr(), C, info["out_io_min"], B, 0)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.blocks): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.blocks
        print(f"lane_hw {args.lane_hw} fmt {fmt} {case:5s}: {C} ch x ({S} sections + {T} taps), block {B}: {ms:9.3f} ms per block = {C * B / ms / 1e6:8.3f} Gsamples/s", flush=True)
        r.release()
