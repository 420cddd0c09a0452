#!/usr/bin/env python3
"""Where a fir_tile wave's time goes: builds the library with -DAVDSP_FIR_STAMPS into /tmp (s_memtime stamps around every chunk's
staging and MFMA phases, per wave, plus the SIMD the wave ran on), runs a few blocks of a workload and summarises the last launch.
    python tools/fir_timeline.py [north|cfg4|...] [--shard r/N] [--fir-rows R]        (on the GPU box)"""
import argparse
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("workload", nargs="?", default="north")
ap.add_argument("--shard", default=None)
ap.add_argument("--fir-rows", type=int, default=0)
ap.add_argument("--fir-impl", type=int, default=3)
ap.add_argument("--fir-split", type=int, default=0)
ap.add_argument("--blocks", type=int, default=6, help="blocks run before the one that is summarised")
ap.add_argument("--overlap", type=int, default=-1, help="the library's \"overlap\" option (-1: its default)")
ap.add_argument("--fir-lean", type=int, default=-1)
ap.add_argument("--block", type=int, default=0, help="frames per block call (0: the workload's own)")
ap.add_argument("--no-build", action="store_true", help="reuse /tmp/libavdsp_stamps.so of an earlier call in this session")
args = ap.parse_args()

lib = "/tmp/libavdsp_stamps.so"
src = os.path.join(ROOT, "avdsp_amd", "csrc")
if not (args.no_build and os.path.exists(lib)):
  subprocess.check_call(f"cd {src} && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -DAVDSP_FIR_STAMPS "
                      f"-I../../include -c -o /tmp/k_stamps.o avdsp_kernels.hip && gcc -O2 -std=gnu99 -fPIC -I../../include -c -o /tmp/h_stamps.o avdsp_host.c && "
                      f"gcc -O2 -std=gnu99 -fPIC -I../../include -c -o /tmp/q_stamps.o avdsp_qformat.c && "
                      f"/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -Wl,-rpath,/opt/rocm/lib -o {lib} /tmp/h_stamps.o /tmp/q_stamps.o /tmp/k_stamps.o", shell=True)
os.environ["AVDSP_LIB"] = lib
import torch                                                       # noqa: E402
from avdsp_amd import progbuilder as pb                            # noqa: E402
from avdsp_amd import devmem as dm
from avdsp_amd import runtime as rt                                # noqa: E402
import bench                                                       # noqa: E402

fmt, Cn, S, T, B = bench.WORKLOADS[args.workload]
if args.block > 0:
    B = args.block
r = rt.Runtime(fmt, pb.synth_program(fmt, Cn, S, T))
r.set_option("fir_rows", args.fir_rows)
r.set_option("fir_impl", args.fir_impl)
r.set_option("fir_split", args.fir_split)
if args.overlap >= 0: r.set_option("overlap", args.overlap)
if args.fir_lean >= 0: r.set_option("fir_lean", args.fir_lean)
if args.shard:
    a, b = (int(v) for v in args.shard.split("/"))
    r.set_shard(a, b)
info = r.shard_info()
Cl = info["nchains"]
x = dm.to_device(np.ascontiguousarray(pb.lcg_input(B, Cn, fmt == 6)[:, info["in_io_min"] - Cn:info["in_io_min"] - Cn + Cl]))
y = torch.zeros((B, Cl), dtype=x.dtype, device="cuda")
for _ in range(args.blocks):
    r.run_block_device(x.data_ptr(), Cl, info["in_io_min"], y.data_ptr(), Cl, info["out_io_min"], B, 0)
torch.cuda.synchronize()
L = rt.lib()
nw = 8192 * 4
buf = np.zeros((nw, 32), dtype=np.uint64)
L.avdsp_hip_debug_fir_stamps.argtypes = [C.c_void_p, C.c_int]
n = L.avdsp_hip_debug_fir_stamps(buf.ctypes.data, nw)
st = buf[:n]
live = st[:, 0] != 0
st = st[live].astype(np.int64)
t0 = st[:, 0].min()
print(f"{args.workload} shard {args.shard}: {Cl} chains, {live.sum()} waves ran (s_memtime taken as 2.4 GHz)")
nch = int(((st[:, 1:22] != 0).sum(axis=1).max()) // 3)
stage = np.zeros(len(st)); mfma = np.zeros(len(st))
for c in range(nch):
    b, m0, m1 = st[:, 1 + 3 * c], st[:, 2 + 3 * c], st[:, 3 + 3 * c]
    ok = m1 != 0
    stage[ok] += (m0 - b)[ok]; mfma[ok] += (m1 - m0)[ok]
    print(f"  chunk {c}: staging {np.median((m0 - b)[ok]):8.0f} cycles (p90 {np.percentile((m0 - b)[ok], 90):8.0f})   k-steps {np.median((m1 - m0)[ok]):8.0f} cycles (p90 {np.percentile((m1 - m0)[ok], 90):8.0f})")
if args.fir_impl == 4:
    for c in range(nch - 1):
        e, b = st[:, 3 + 3 * c], st[:, 4 + 3 * c]
        ok = b != 0
        print(f"  chunk {c} -> {c + 1}: the window copy's requests {np.median((b - e)[ok]):6.0f} cycles (p90 {np.percentile((b - e)[ok], 90):6.0f})")
    ok = st[:, 23] != 0
    print("  start -> first chunk's wait %.0f cycles;  epilogue (convert and store the tile) %.0f cycles (p90 %.0f)"
          % (np.median(st[ok, 1] - st[ok, 0]), np.median(st[ok, 30] - st[ok, 23]), np.percentile(st[ok, 30] - st[ok, 23], 90)))
if args.fir_impl == 3 and (st[:, 26] != 0).any():
    ok = st[:, 26] != 0
    print("  first unit: start -> unit known %.0f cycles, -> first chunk's operands ready %.0f;  epilogue (convert and store the tile) %.0f cycles"
          % (np.median(st[ok, 24] - st[ok, 0]), np.median(st[ok, 2] - st[ok, 0]), np.median(st[ok, 26] - st[ok, 25])))
if args.fir_impl == 1 and (st[:, 23] != 0).any():
    ok = st[:, 23] != 0
    print("  start -> first chunk's boundary %.0f cycles;  epilogue (convert and store the tile) %.0f cycles (p90 %.0f)"
          % (np.median(st[ok, 1] - st[ok, 0]), np.median(st[ok, 30] - st[ok, 23]), np.percentile(st[ok, 30] - st[ok, 23], 90)))
if args.fir_impl not in (3, 4) and (st[:, 24] != 0).any():
    ok = st[:, 27] != 0
    b = st[ok, 4]                                    # start of chunk 1's boundary
    print("  chunk 1's boundary: requested data landed +%.0f, window image written +%.0f, taps image landed +%.0f, next window requested +%.0f, next taps requested +%.0f, first operands read +%.0f cycles"
          % tuple(np.median(st[ok, i] - b) for i in (4, 24, 25, 26, 27, 5)))
life = st[:, 30] - st[:, 0]
rt_ = (st[:, 28] - st[:, 29]).astype(np.float64)
okc = rt_ > 0
clk = life[okc] / rt_[okc] * 100e6
print(f"  in-kernel clock (s_memtime / s_memrealtime x 100 MHz over a wave's life): median {np.median(clk) / 1e9:.3f} GHz, p10 {np.percentile(clk, 10) / 1e9:.3f}, p90 {np.percentile(clk, 90) / 1e9:.3f};  wave life by the 100 MHz clock: median {np.median(rt_[okc]) / 100:.1f} us")
print(f"  per wave: life {np.median(life) / 2400:.1f} us, staging {np.median(stage) / 2400:.1f} us ({100 * np.median(stage / life):.1f} %), k-steps {np.median(mfma) / 2400:.1f} us, "
      f"start spread {(st[:, 0].max() - t0) / 2400:.1f} us, end spread {(st[:, 30].max() - st[:, 30].min()) / 2400:.1f} us")
rs = (st[:, 29] - st[:, 29].min()) / 100.0; re_ = (st[:, 28] - st[:, 29].min()) / 100.0        # s_memrealtime: 100 MHz, one clock for the chip
print("  wave starts (us after the first, by the 100 MHz clock): p10 %.1f p50 %.1f p90 %.1f max %.1f;  ends: p10 %.1f p50 %.1f p90 %.1f max %.1f"
      % (*np.percentile(rs, [10, 50, 90, 100]), *np.percentile(re_, [10, 50, 90, 100])))
hist, edges = np.histogram(rs, bins=12)
print("  wave starts, histogram over the launch: " + " ".join(f"{int(e)}us:{h}" for h, e in zip(hist, edges[:-1])))
hw = (st[:, 31].astype(np.uint64) >> np.uint64(32)).astype(np.int64)
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; xcc = st[:, 31] & 7
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
key = key * 4 + simd
uniq, cnt = np.unique(key, return_counts=True)
print(f"  SIMDs used {len(uniq)}, waves per SIMD over the launch: min {cnt.min()} median {int(np.median(cnt))} max {cnt.max()}")
# MFMA time a SIMD would need if its waves never waited for each other vs the span
per_simd_busy = np.array([mfma[key == u].sum() for u in uniq])
print(f"  sum of k-step time per SIMD: median {np.median(per_simd_busy) / 2400:.1f} us, max {per_simd_busy.max() / 2400:.1f} us")
r.release()
