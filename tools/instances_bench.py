#!/usr/bin/env python3
"""N instances of one of the reference's own programs side by side (dspRuntimeSetInstances): Gsamples/s over all instances, input
channels counted, blocks resident in HBM, and the check of a few instances against the golden run of ONE instance of the same input.
    python tools/instances_bench.py [--prog crossoverLV6.bin] [--instances 1 64 1024 4096] [--frames 4096]      (on the GPU box)"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from avdsp_amd import progbuilder as pb      # noqa: E402
from avdsp_amd import devmem as dm
from avdsp_amd import runtime as rt          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--prog", default="crossoverLV6.bin")
ap.add_argument("--instances", type=int, nargs="*", default=[1, 64, 256, 1024, 4096])
ap.add_argument("--frames", type=int, default=4096)
ap.add_argument("--blocks", type=int, default=10)
ap.add_argument("--windows", type=int, nargs=4, default=None, metavar=("IN_STRIDE", "IN_BASE", "OUT_STRIDE", "OUT_BASE"),
                help="the call's windows (default: crossoverLV6's own IOs, 2 16 8 24, which share no IO number; 16 8 32 0 are the goldens' windows, "
                     "which do -- whole rows move then and the pieces of a level run one after the other: dacdiy1.bin needs that)")
ap.add_argument("--synth", type=int, nargs=4, default=None, metavar=("FMT", "CH", "SECTIONS", "TAPS"),
                help="instead of a program file: a chain program as bench.py builds them (cfg2 = 6 8 8 0) -- chain cores in instances (round 5)")
args = ap.parse_args()
if args.synth:
    fmt, Cc, S, T = args.synth
    prog = pb.synth_program(fmt, Cc, S, T)
    B = args.frames
    x1 = pb.lcg_input(B, Cc, fmt in (5, 6), seed=5)
    r0 = rt.Runtime(fmt, prog)
    want = r0.run_block(x1, Cc, Cc)
    r0.release()
    for n in args.instances:
        r = rt.Runtime(fmt, prog)
        r.set_instances(n)
        x = dm.to_device(x1).unsqueeze(0).repeat(n, 1, 1).contiguous()
        y = torch.zeros((n, B, Cc), dtype=x.dtype, device="cuda")
        st = torch.cuda.current_stream().cuda_stream
        run = lambda: r.run_block_all_instances_device(x.data_ptr(), Cc, Cc, B * Cc, y.data_ptr(), Cc, 0, B * Cc, B, st)
        run(); torch.cuda.synchronize()
        got = dm.to_host(y)
        ok = all((got[i].view(np.uint32) == want.view(np.uint32)).all() for i in sorted({0, n // 2, n - 1}))
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.blocks):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.blocks
        print(f"synth fmt {fmt}: {Cc} ch x ({S} biquads + {T} taps): {n:5d} instances x {B} frames: {dt * 1e6:9.1f} us per block, "
              f"{n * B * Cc / dt / 1e9:7.2f} Gsamples/s over all instances ({n * Cc} chains in one launch set); first block of instances 0, n/2, n-1 == one instance alone: {ok}", flush=True)
        r.release()
    sys.exit(0)
prog = np.fromfile(os.path.join(ROOT, "tests", "golden", args.prog), dtype=np.uint32)
IN_STRIDE, IN_BASE, OUT_STRIDE, OUT_BASE = args.windows or (2, 16, 8, 24)      # crossoverLV6: inputs IO 16, 17, outputs IO 25 .. 29 (windows that share no IO number)
used_in = bin(int(prog[9])).count("1")              # header.usedInputs (word 9): the channels the program really reads
B = args.frames
x1 = pb.lcg_input(B, IN_STRIDE, False, seed=5)
# one instance through the ordinary entry point: what every instance of the same input must give
r0 = rt.Runtime(2, prog, fs=48000, random=3, dither=24)
want = r0.run_block_all(x1, OUT_STRIDE, IN_BASE, OUT_BASE)
r0.release()
for n in args.instances:
    r = rt.Runtime(2, prog, fs=48000, random=3, dither=24)
    r.set_instances(n)
    x = dm.to_device(x1).unsqueeze(0).repeat(n, 1, 1).contiguous()
    y = torch.zeros((n, B, OUT_STRIDE), dtype=x.dtype, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    run = lambda: r.run_block_all_instances_device(x.data_ptr(), IN_STRIDE, IN_BASE, B * IN_STRIDE, y.data_ptr(), OUT_STRIDE, OUT_BASE, B * OUT_STRIDE, B, st)
    run(); torch.cuda.synchronize()
    got = dm.to_host(y)
    ok = all((got[i].view(np.uint32) == want.view(np.uint32)).all() for i in sorted({0, n // 2, n - 1}))
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.blocks):
        run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.blocks
    print(f"{args.prog}: {n:5d} instances x {B} frames: {dt * 1e3:8.3f} ms per block = {dt / B * 1e6:7.3f} us per frame of all instances, "
          f"{n * B * used_in / dt / 1e9:7.3f} Gsamples/s ({used_in} input channels each; {n * B / dt / 1e6:8.1f} M instance-frames/s = "
          f"{n * B / dt / 48000:7.0f} x real time at 48 kHz in all); first block of instances 0, n/2, n-1 == one instance alone: {ok}", flush=True)
    r.release()
