"""Development probe: dspRuntimeBlockAllDevice on a side stream / with profiling; prints its stages.
usage: python tests/dev/gpu_all_diag.py PROFILE(0|1) SIDE(0|1)"""
import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
from avdsp_amd import progbuilder as pb, runtime as rt
profile, use_side = int(sys.argv[1]), int(sys.argv[2])
prog = np.fromfile("tests/golden/dacdiy1.bin", dtype=np.uint32)
xh = pb.lcg_input(512, 16, False, seed=5)
r = rt.Runtime(2, prog, fs=48000, random=3, dither=24)
print("runtime", flush=True)
r.set_option("profile", profile)
stream = torch.cuda.Stream() if use_side else torch.cuda.current_stream()
x = torch.from_numpy(xh).cuda(); y = torch.zeros((512, 8), dtype=torch.int32, device="cuda")
torch.cuda.synchronize(); print("buffers", flush=True)
with torch.cuda.stream(stream):
    for b0 in (0, 256):
        rc = r.L.dspRuntimeBlockAllDevice(2, r.rundata, x[b0:].data_ptr(), 16, 8, y[b0:].data_ptr(), 8, 0, 256, stream.cuda_stream)
        print("enqueued", b0, rc, flush=True)
stream.synchronize(); print("stream done", flush=True)
torch.cuda.synchronize(); print("device done", flush=True)
print("state", r.sync_state()[:4], flush=True)
if profile: print("kernel_time", r.kernel_time(5), r.kernel_time(3), flush=True)
r.release(); print("released", flush=True)
