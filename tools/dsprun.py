#!/usr/bin/env python3
"""A dsprun-like file runner on the GPU path (the reference's linux/dsprun.c:60-176 is the model):
load an encoded program, find its cores and their IO maps from the DSP_CORE words, feed a stimulus
(impulse / 40 Hz sine / noise, as dsprun's -i / -s / -r, or a raw interleaved PCM file), run it in
blocks through libavdsp_mi355x.so and write the outputs as a 32-bit WAV or raw file.

    python tools/dsprun.py -i out.wav tests/golden/crossoverLV6.bin 48000 [--frames N] [--block B]
    python tools/dsprun.py --raw in.s24 --pcm s24_3le --channels 2 out.raw prog.bin 96000

Int-sample programs (DSP_FORMAT 2 by default, --format 3/4 for float-encoded ones)."""
import argparse
import os
import sys
import wave

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avdsp_amd import runtime as rt      # noqa: E402

def io_maps(r):
    """usedInputs / usedOutputs of every DSP_CORE (dsprun.c:103-131, with all 32 mask bits instead of 16):
    returns (first input IO, input channels, first output IO, output channels)."""
    ins, outs = 0, 0
    base = r.buf.ctypes.data
    k = 1
    while True:
        p = r.L.dspFindCore(base, k)
        if not p:
            break
        at = (p - base) // 4
        ins |= int(r.buf[at + 1]); outs |= int(r.buf[at + 2])
        k += 1
    i = [ch for ch in range(32) if ins >> ch & 1] or [0]
    o = [ch for ch in range(32) if outs >> ch & 1] or [0]
    return i[0], i[-1] - i[0] + 1, o[0], o[-1] - o[0] + 1


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    g = ap.add_mutually_exclusive_group(required=True)
    g.add_argument("-i", dest="mode", action="store_const", const="impulse")
    g.add_argument("-s", dest="mode", action="store_const", const="sine")
    g.add_argument("-r", dest="mode", action="store_const", const="noise")
    g.add_argument("--raw", metavar="FILE")
    ap.add_argument("out"); ap.add_argument("program"); ap.add_argument("fs", type=int)
    ap.add_argument("--frames", type=int, default=0, help="default: 5 s (dsprun.c:145)")
    ap.add_argument("--block", type=int, default=4096)
    ap.add_argument("--format", type=int, default=2, choices=(2, 3, 4))
    ap.add_argument("--pcm", default="s32", choices=("s32", "s24_3le", "s16"))
    ap.add_argument("--channels", type=int, default=0, help="channels in the raw input (default: what the program reads)")
    ap.add_argument("--dither", type=int, default=31)
    a = ap.parse_args()

    prog = np.fromfile(a.program, dtype=np.uint32)
    r = rt.Runtime(a.format, prog, fs=a.fs, random=0, dither=a.dither)
    if r.rc < 0:
        sys.exit(f"dspRuntimeInit: {r.rc} ({r.last_error()})")
    in_base, nin, out_base, nout = io_maps(r)
    nin = a.channels or nin
    pcm = {"s32": rt.PCM_S32, "s24_3le": rt.PCM_S24_3LE, "s16": rt.PCM_S16}[a.pcm]
    width = {rt.PCM_S32: 4, rt.PCM_S24_3LE: 3, rt.PCM_S16: 2}[pcm]
    if a.raw:
        raw = np.fromfile(a.raw, dtype=np.uint8)
        frames = raw.size // (width * nin)
        raw = raw[:frames * width * nin]
    else:
        frames = a.frames or 5 * a.fs
        n = np.arange(frames)
        if a.mode == "impulse":
            col = np.zeros(frames, dtype=np.int64); col[0] = 2**31 - 1
        elif a.mode == "sine":
            col = np.round((2**31 - 1) * np.sin(2 * np.pi * 40.0 * n / a.fs)).astype(np.int64)
        else:
            col = np.random.default_rng(1).integers(-2**27, 2**27, frames)
        x = np.repeat(col.astype(np.int32)[:, None], nin, axis=1)
        raw, pcm = np.ascontiguousarray(x).view(np.uint8).reshape(-1), rt.PCM_S32
    if a.frames:
        frames = min(frames, a.frames)
    out = r.run_block_all_pcm(pcm, raw[:frames * nin * {rt.PCM_S32: 4, rt.PCM_S24_3LE: 3, rt.PCM_S16: 2}[pcm]], frames, nin, nout,
                              in_base, out_base, block=a.block)          # every core per block, cores that do not meet together
    if a.out.endswith(".wav"):
        with wave.open(a.out, "wb") as w:
            w.setnchannels(nout); w.setsampwidth(4); w.setframerate(a.fs)
            w.writeframes(out.astype("<i4").tobytes())
    else:
        out.astype("<i4").tofile(a.out)
    print(f"{len(r.cores)} core(s), inputs IO {in_base}..{in_base + nin - 1}, outputs IO {out_base}..{out_base + nout - 1}, "
          f"{frames} frames at {a.fs} Hz -> {a.out}")


if __name__ == "__main__":
    main()
