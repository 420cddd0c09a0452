#!/usr/bin/env python3
"""Where a biquad_pipe wave's time goes: builds the library with -DAVDSP_BQ_STAMPS into /tmp (s_memtime at the kernel's start,
in front of the batch loop, at the head of batches 12..35 and behind the loop), runs a few blocks and summarises the last launch.
    python tools/cascade_timeline.py [cfg3|north|...] [--shard r/N] [--blocks N]        (on the GPU box)"""
import argparse
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("workload", nargs="?", default="cfg3")
ap.add_argument("--shard", default=None)
ap.add_argument("--blocks", type=int, default=300)
ap.add_argument("--biquad-impl", type=int, default=1)
ap.add_argument("--block", type=int, default=0, help="frames per block call (0: the workload's own)")
args = ap.parse_args()
lib = "/tmp/libavdsp_bqstamps.so"
src = os.path.join(ROOT, "avdsp_amd", "csrc")
subprocess.check_call(f"cd {src} && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -DAVDSP_BQ_STAMPS "
                      f"-I../../include -c -o /tmp/k_bqs.o avdsp_kernels.hip && gcc -O2 -std=gnu99 -fPIC -I../../include -c -o /tmp/h_bqs.o avdsp_host.c && "
                      f"gcc -O2 -std=gnu99 -fPIC -I../../include -c -o /tmp/q_bqs.o avdsp_qformat.c && "
                      f"/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -Wl,-rpath,/opt/rocm/lib -o {lib} /tmp/h_bqs.o /tmp/q_bqs.o /tmp/k_bqs.o", shell=True)
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
r.set_option("biquad_impl", args.biquad_impl)
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
nw = 65536 * 4
buf = np.zeros((nw, 32), dtype=np.uint64)
L.avdsp_hip_debug_bq_stamps.argtypes = [C.c_void_p, C.c_int]
n = L.avdsp_hip_debug_bq_stamps(buf.ctypes.data, nw)
st = buf[:n]
st = st[st[:, 29] != 0].astype(np.int64)
print(f"{args.workload} shard {args.shard}: {Cl} chains x {S} sections, {len(st)} waves")
med = lambda v: float(np.median(v))
print(f"  start -> batch loop (coefficients, state, first samples): {med(st[:, 1] - st[:, 0]):.0f} cycles")
nbat = (B + 2 * (S - 1) + 1 + 15) // 16
per = np.diff(st[:, 2:2 + max(2, min(24, nbat - 12 - 6))], axis=1) if nbat > 20 else np.zeros((1, 1))
print(f"  one batch of 16 steps, batches 12..34: median {med(per):.0f} cycles (p10 {np.percentile(per, 10):.0f}, p90 {np.percentile(per, 90):.0f}) = {med(per) / 16:.1f} per step")
print(f"  batch loop in all: {med(st[:, 28] - st[:, 1]):.0f} cycles;  behind the loop (state write-back, Inf/NaN look): {med(st[:, 29] - st[:, 28]):.0f} cycles")
if (st[:, 26] != 0).any():          # biquad_row: the fill batches, the loop of steady batches, the rest (steady leftovers + drain)
    print(f"  fill batches {med(st[:, 26] - st[:, 1]):.0f} cycles, steady loop {med(st[:, 27] - st[:, 26]):.0f}, leftover + drain batches {med(st[:, 28] - st[:, 27]):.0f}")
print(f"  wave life {med(st[:, 29] - st[:, 0]):.0f} cycles")
r.release()
