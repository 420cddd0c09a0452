#!/usr/bin/env python3
"""What the operations of a strand cost strand_lanes per frame: programs of 128 strands that grow by one opcode at a time
(kernel time of strand_lanes from the library's timers).  Run on the GPU box."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STAMPS = "--stamps" in sys.argv          # build with -DAVDSP_BQ_STAMPS into /tmp: s_memtime in front of every operation of batch 8
if STAMPS:
    lib = "/tmp/libavdsp_bqstamps.so"
    src = os.path.join(ROOT, "avdsp_amd", "csrc")
    subprocess.check_call(f"cd {src} && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 -DAVDSP_BQ_STAMPS "
                          f"-I../../include -c -o /tmp/k_bqs.o avdsp_kernels.hip && gcc -O2 -std=gnu99 -fPIC -I../../include -c -o /tmp/h_bqs.o avdsp_host.c && "
                          f"gcc -O2 -std=gnu99 -fPIC -I../../include -c -o /tmp/q_bqs.o avdsp_qformat.c && "
                          f"/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -Wl,-rpath,/opt/rocm/lib -o {lib} /tmp/h_bqs.o /tmp/q_bqs.o /tmp/k_bqs.o", shell=True)
    os.environ["AVDSP_LIB"] = lib
import ctypes as C
import numpy as np
import torch
from avdsp_amd import encoder as enc, progbuilder as pb, runtime as rt
from avdsp_amd import devmem as dm
from tests.fuzz_programs import _prototypes

FPEAK, F48000 = 74, 5
STEPS = ["load_store", "+gain", "+biquads2", "+delay", "+sat_tpdf"]


def program(nch, fmt, upto):
    def build(L):
        banks = []
        for c in range(nch):
            if c % 64 == 0:
                L.dsp_PARAM()
            b = L.dspBiquad_Sections(2)
            for k in range(2):
                L.dsp_Filter2ndOrder(FPEAK, 150.0 * (k + 1) + 7 * c, 1.0, 0.95)
            banks.append(b)
        L.dsp_CORE()
        if upto >= 4:
            L.dsp_TPDF_CALC(0)
        for c in range(nch):
            L.dsp_LOAD_GAIN_Fixed(nch + c, 0.5)
            if upto >= 1: L.dsp_GAIN_Fixed(0.9)
            if upto >= 2: L.dsp_BIQUADS(banks[c])
            if upto >= 3: L.dsp_DELAY_FixedMicroSec(100 + 10 * (c % 40))
            if upto >= 4: L.dsp_SAT0DB_TPDF()
            L.dsp_STORE(c)
    L = enc.lib(); _prototypes(L)
    return enc.encode(build, 2 if fmt == 2 else 6, F48000, F48000, max_io=2 * nch + 8, capacity=1 << 17)


nch, frames = 128, 4096
for fmt in (2, 6):
    x = dm.to_device(pb.lcg_input(frames, nch, fmt == 6, seed=1))
    y = torch.zeros((frames, nch), dtype=x.dtype, device="cuda")
    prev = 0.0
    for upto, name in enumerate(STEPS):
        r = rt.Runtime(fmt, program(nch, fmt, upto), fs=48000, random=1, dither=24)
        r.set_option("strand_lanes", 2)
        st = torch.cuda.current_stream().cuda_stream
        call = lambda: r._check(r.L.dspRuntimeBlockAllDevice(fmt, r.rundata, x.data_ptr(), nch, nch, y.data_ptr(), nch, 0, frames, st))
        for _ in range(3): call()
        torch.cuda.synchronize()
        r.set_option("profile", 1)
        r.kernel_time(6)
        for _ in range(5): call()
        torch.cuda.synchronize()
        ms, n = r.kernel_time(6)
        us = ms * 1e3 / max(n, 1) / frames
        print(f"fmt {fmt} {name:12s}: strands {r.get_option('strands'):4d}  strand_lanes {us:6.3f} us/frame (+{us - prev:6.3f})", flush=True)
        prev = us
        r.set_option("profile", 0)
        if STAMPS and r.get_option("strands"):
            buf = np.zeros(32, dtype=np.uint64)
            r.L.avdsp_hip_debug_bq_stamps.argtypes = [C.c_void_p, C.c_int]
            if r.L.avdsp_hip_debug_bq_stamps(buf.ctypes.data, 1) == 1:
                t = buf.astype(np.int64); k = int(np.count_nonzero(t))
                print("      cycles per operation of one batch of 16 frames (s_memtime): " + " ".join(str(int(v)) for v in np.diff(t[:k])), flush=True)
        r.set_option("strand_lanes", 1)
        r.L.dspRuntimeRelease()
