#!/usr/bin/env python3
"""Side measurements for DESIGN.md (not the bench.py metric): the PCM unpack kernel against the HBM
roofline, and the general interpreter's frame rate on the reference's committed programs.
Run on the GPU box:  python tools/extras_bench.py"""
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

HBM_GBS = 8000.0


def unpack():
    r = rt.Runtime(2, pb.synth_program(2, 2, 1))
    r.run_block(np.zeros((1, 2), dtype=np.int32), 2, 2)
    r.set_option("profile", 1)
    for name, pcm, width in (("S16", rt.PCM_S16, 2), ("S24_3LE", rt.PCM_S24_3LE, 3)):
        for n in (4096 * 1024, 64 * 1024 * 1024):
            src = torch.randint(0, 255, (n * width,), dtype=torch.uint8, device="cuda")
            dst = torch.empty(n, dtype=torch.int32, device="cuda")
            for _ in range(3):
                r.L.dspRuntimeUnpackPcmDevice(pcm, src.data_ptr(), dst.data_ptr(), n, None)
            torch.cuda.synchronize()
            r.kernel_time(4)
            reps = 20
            for _ in range(reps):
                r.L.dspRuntimeUnpackPcmDevice(pcm, src.data_ptr(), dst.data_ptr(), n, None)
            torch.cuda.synchronize()
            ms, k = r.kernel_time(4)
            us = ms * 1e3 / k
            gbs = n * (width + 4) / (us * 1e-6) / 1e9
            print(f"unpack {name:8s} {n:>9d} samples: {us:8.1f} us/launch  {gbs:7.0f} GB/s  frac {gbs / HBM_GBS:.3f}")
    r.L.dspRuntimeRelease()


def interpreter():
    gold = os.path.join(ROOT, "tests", "golden")
    for name, fmt, frames in (("crossoverLV6.bin", 2, 48000), ("dacdiy1.bin", 2, 48000), ("tour_float.bin", 6, 48000),
                              ("tour_float.bin", 3, 48000)):
        prog = np.fromfile(os.path.join(gold, name), dtype=np.uint32)
        tour = name.startswith("tour")
        x = pb.lcg_input(frames, 16, fmt in (5, 6), seed=5)
        in_base, out_stride = (32, 32) if tour else (8, 32)
        for impl, label in ((1, "frame-parallel"), (0, "frame by frame")):
            r = rt.Runtime(fmt, prog, fs=48000, random=1, dither=24)
            r.set_option("interp_impl", impl)
            r.run_block(x[:64], out_stride, in_base)                         # plans, staging buffers
            for _ in range(8):                                               # the chip raises its clock over the first ~0.1 s of load
                r.run_block(x, out_stride, in_base)
            r.set_option("profile", 1)
            r.kernel_time(3); r.kernel_time(5)
            t0 = time.perf_counter()
            r.run_block(x, out_stride, in_base)
            wall = time.perf_counter() - t0
            ms3, k3 = r.kernel_time(3)
            ms5, k5 = r.kernel_time(5)
            ms = ms3 + ms5
            print(f"interp {name:18s} fmt {fmt} {label:15s}: {len(r.cores)} cores, {frames} frames: kernels {ms:8.2f} ms "
                  f"({ms * 1e3 / frames:6.3f} us/frame, {frames / (ms * 1e-3) / 48000:7.1f}x real time at 48 kHz; "
                  f"launches frame-parallel {k5} / frame by frame {k3}), wall {wall * 1e3:.1f} ms")
            r.set_option("profile", 0)
            r.set_option("interp_impl", 1)
            r.L.dspRuntimeRelease()


def program_level():
    """Whole programs on device-resident blocks, stream time (HIP events): the cores one after the other
    (dspRuntimeBlockDevice per core) against dspRuntimeBlockAllDevice (cores that do not meet side by side)."""
    gold = os.path.join(ROOT, "tests", "golden")
    # windows that do not share IO numbers (with shared ones the whole rows are delivered and cores run in turn)
    for name, fmt, in_stride, in_base, out_stride in (("crossoverLV6.bin", 2, 16, 8, 8), ("dacdiy1.bin", 2, 16, 8, 8),
                                                      ("tour_float.bin", 6, 16, 32, 32)):
        prog = np.fromfile(os.path.join(gold, name), dtype=np.uint32)
        for frames in (256, 4096):
            r = rt.Runtime(fmt, prog, fs=48000, random=1, dither=24)
            if os.environ.get("STRAND_LANES") is not None:                   # 0: strand runs stay with the interpreter's strand groups
                r.set_option("strand_lanes", int(os.environ["STRAND_LANES"]))
            x = dm.to_device(pb.lcg_input(frames, in_stride, fmt in (5, 6), seed=5))
            y = torch.zeros((frames, out_stride), dtype=x.dtype, device="cuda")
            stream = torch.cuda.current_stream().cuda_stream
            res = {}
            for mode in ("per core", "all"):
                def once():
                    if mode == "all":
                        r._check(r.L.dspRuntimeBlockAllDevice(fmt, r.rundata, x.data_ptr(), in_stride, in_base, y.data_ptr(),
                                                              out_stride, 0, frames, stream))
                    else:
                        for k in range(len(r.cores)):
                            r.run_block_device(x.data_ptr(), in_stride, in_base, y.data_ptr(), out_stride, 0, frames, stream, k)
                for _ in range(3):
                    once()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps = 20
                e0.record()
                for _ in range(reps):
                    once()
                e1.record()
                torch.cuda.synchronize()
                res[mode] = e0.elapsed_time(e1) * 1e3 / reps / frames
            r.set_option("profile", 1)
            for k in (3, 5, 6): r.kernel_time(k)
            once(); torch.cuda.synchronize()
            kt = {k: r.kernel_time(k) for k in (3, 5, 6)}
            r.set_option("profile", 0)
            print(f"program {name:18s} fmt {fmt} block {frames:5d}: per core {res['per core']:6.3f} us/frame, "
                  f"BlockAll {res['all']:6.3f} us/frame ({r.get_option('cores')} cores in {r.get_option('levels')} levels, {r.get_option('pieces')} pieces; "
                  f"launches of one call: frame-parallel {kt[5][1]} = {kt[5][0] * 1e3 / frames:.3f} us/frame, frame by frame {kt[3][1]}, strand plans {kt[6][1]})", flush=True)
            r.set_option("strand_lanes", 1)
            r.L.dspRuntimeRelease()


def single_frame():
    """dspRuntime_N, the reference's own entry point: one frame per call, samples[] in host memory (BASELINE config 1 is
    "block = 1").  Every call is a PCIe round trip and one kernel launch per core; microseconds per call, wall clock."""
    gold = os.path.join(ROOT, "tests", "golden")
    for name, fmt in (("crossoverLV6.bin", 2), ("dacdiy1.bin", 2)):
        prog = np.fromfile(os.path.join(gold, name), dtype=np.uint32)
        r = rt.Runtime(fmt, prog, fs=48000, random=1, dither=24)
        frame = np.zeros(64, dtype=np.int32)
        x = pb.lcg_input(1200, 16, False, seed=5)
        lat = []
        for n in range(1200):
            frame[8:24] = x[n]
            t0 = time.perf_counter()
            for k in range(len(r.cores)):
                r.run_frame(frame, k)
            lat.append(time.perf_counter() - t0)
        lat = np.array(lat[200:]) * 1e6
        print(f"single frame {name:18s} fmt {fmt}: dspRuntime_{fmt} over {len(r.cores)} cores, 1000 frames: median {np.median(lat):7.1f} us per frame "
              f"(p10 {np.percentile(lat, 10):.1f}, p90 {np.percentile(lat, 90):.1f}) = {np.median(lat) / len(r.cores):.1f} us per core call; "
              f"real time at 48 kHz allows 20.8 us per frame", flush=True)
        r.L.dspRuntimeRelease()


if __name__ == "__main__":
    if "single" in sys.argv[1:]:
        single_frame()
        sys.exit(0)
    if "program" in sys.argv[1:]:
        program_level()
        sys.exit(0)
    if "interp" not in sys.argv[1:]:
        unpack()
    interpreter()
