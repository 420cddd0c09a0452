#!/usr/bin/env python3
"""What an event pair around every launch costs the stream: the north-star step (biquad + FIR kernel) with the
library's per-kernel HIP events on and off, wall time per step over 30 steps.  Run on the GPU box:
python tools/step_gap_probe.py   ->  profile 1: 0.62 ms/step, profile 0: 0.60 ms/step"""
import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from avdsp_amd import progbuilder as pb, runtime as rt, sharding as sh
from avdsp_amd import devmem as dm
C,S,T,B=4096,16,4096,1024
prog,lo,hi=sh.shard_program(6,C,S,T,1,0)
for prof in (1,0,1,0):
    r=rt.Runtime(6,prog); r.set_option("profile",prof)
    x=dm.to_device(pb.lcg_input(B,C,True,seed=12345)); y=torch.zeros((B,C),dtype=x.dtype,device="cuda")
    st=torch.cuda.current_stream().cuda_stream
    for _ in range(5): r.run_block_device(x.data_ptr(),C,C,y.data_ptr(),C,0,B,st)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(30): r.run_block_device(x.data_ptr(),C,C,y.data_ptr(),C,0,B,st)
    torch.cuda.synchronize(); el=time.perf_counter()-t0
    print("profile",prof,"ms/step",el/30*1e3, flush=True)
    r.release()
