#!/usr/bin/env python3
"""fir_mfma launch time against the number of channels (= workgroups) at 4096 taps, B = 1024: separates the
per-round cost from ramp-up / tail effects.  GPU box: python tools/fir_scaling.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from avdsp_amd import progbuilder as pb, runtime as rt
from avdsp_amd import devmem as dm

T, B = int(os.environ.get("TAPS", 4096)), 1024
taps = pb.lcg_taps_all(1, T)
for C in (256, 512, 1024, 1536, 2048, 3072, 4096, 6144, 8192):
    prog = pb.synth_program(6, C, 0, T, shared_taps=True, taps=taps)
    r = rt.Runtime(6, prog)
    r.set_option("profile", 1)
    x = dm.to_device(pb.lcg_input(B, C, True))
    y = torch.zeros((B, C), dtype=torch.float32, device="cuda")
    for _ in range(3):
        r.run_block_device(x.data_ptr(), C, C, y.data_ptr(), C, 0, B)
    torch.cuda.synchronize(); r.kernel_time(1)
    for _ in range(10):
        r.run_block_device(x.data_ptr(), C, C, y.data_ptr(), C, 0, B)
    torch.cuda.synchronize()
    ms, n = r.kernel_time(1)
    us = ms * 1e3 / n
    print(f"C={C:5d} blocks/CU={C/256:5.1f}  {us:8.1f} us  {us/(C/1024):7.1f} us per 1024 channels  {2.0*T*B*C/(us*1e-6)/1e12:6.2f} TFLOP/s")
    r.release()
