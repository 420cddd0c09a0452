#!/usr/bin/env python3
"""Quick on-GPU diagnostic: parity of each kernel vs the oracle (bit-exact or max error) and rough
timings of the BASELINE configs.  Prints, never asserts; meant for gpurun while developing."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from avdsp_amd import progbuilder as pb, runtime as rt
from oracle import pyoracle as po


def cmp(name, fmt, prog, x, C, opts=None, blocks=None):
    o = po.OracleProgram(fmt, prog)
    r = rt.Runtime(fmt, prog)
    for k, v in (opts or {}).items():
        r.set_option(k, v)
    try:
        blocks = blocks or [len(x)]
        pos = 0; got = []; want = []
        for b in blocks:
            want.append(o.run_block(x[pos:pos + b], C, C)); got.append(r.run_block(x[pos:pos + b], C, C)); pos += b
        got = np.concatenate(got); want = np.concatenate(want)
        nd = np.count_nonzero(got.view(np.uint32) != want.view(np.uint32))
        err = np.abs(got.astype(np.float64) - want.astype(np.float64)).max() / max(np.abs(want.astype(np.float64)).max(), 1e-30)
        st = r.sync_state(); sd = np.count_nonzero(st != o.state)
        print(f"{name:46s} words differing {nd:8d}/{got.size:<8d} rel.err {err:.3e}  state words differing {sd}", flush=True)
        if nd:
            idx = np.argwhere(got.view(np.uint32) != want.view(np.uint32))[:4]
            for i in idx: print("     at", tuple(i), "got", got[tuple(i)], "want", want[tuple(i)])
    except rt.AvdspError as e:
        print(f"{name:46s} ERROR {e}", flush=True)
    r.release()


def bench(name, fmt, C, S, T, B, opts=None, reps=10):
    import torch
    taps = pb.lcg_taps_all(C, T) if T else None
    prog = pb.synth_program(fmt, C, S, T, taps=taps)
    r = rt.Runtime(fmt, prog)
    for k, v in (opts or {}).items():
        r.set_option(k, v)
    x = torch.from_numpy(pb.lcg_input(B, C, fmt == 6)).cuda()
    y = torch.zeros((B, C), dtype=x.dtype, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        r.run_block_device(x.data_ptr(), C, C, y.data_ptr(), C, 0, B, s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        r.run_block_device(x.data_ptr(), C, C, y.data_ptr(), C, 0, B, s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{name:46s} {ms:9.3f} ms/block  {C * B / ms / 1e3:10.1f} Msamples/s", flush=True)
    r.release()


if __name__ == "__main__":
    print("devices:", rt.lib().avdsp_hip_device_count())
    for fmt in (2, 4, 6):
        fl = fmt == 6
        for impl in (0, 1):
            cmp(f"bq 8ch x 8sec fmt{fmt} biquad_impl={impl}", fmt, pb.synth_program(fmt, 8, 8), pb.lcg_input(256, 8, fl), 8, {"biquad_impl": impl})
        cmp(f"bq 37ch x 5sec fmt{fmt} ragged", fmt, pb.synth_program(fmt, 37, 5), pb.lcg_input(158, 37, fl), 37, None, [1, 7, 16, 100, 33, 1])
        cmp(f"bq 3ch x 40sec fmt{fmt}", fmt, pb.synth_program(fmt, 3, 40), pb.lcg_input(100, 3, fl), 3)
    for fmt in (4, 6):
        fl = fmt == 6
        for impl in (0, 1):
            for T in (7, 255, 1025):
                cmp(f"fir 3ch x {T}tap fmt{fmt} fir_impl={impl}", fmt, pb.synth_program(fmt, 3, 0, T), pb.lcg_input(700, 3, fl), 3, {"fir_impl": impl}, [300, 1, 399])
        cmp(f"mixed 6ch 4sec+129tap fmt{fmt}", fmt, pb.synth_program(fmt, 6, 4, 129), pb.lcg_input(600, 6, fl), 6)
    if "--bench" in sys.argv:
        bench("cfg2  8ch x 8bq fmt6 B=256", 6, 8, 8, 0, 256)
        bench("cfg3  4096ch x 16bq fmt2 B=1024", 2, 4096, 16, 0, 1024)
        bench("cfg3  4096ch x 16bq fmt6 B=1024", 6, 4096, 16, 0, 1024)
        bench("cfg3  4096ch x 16bq fmt6 B=1024 simple", 6, 4096, 16, 0, 1024, {"biquad_impl": 0}, reps=3)
        bench("cfg4  256ch x 4096tap fmt6 B=1024 mfma", 6, 256, 0, 4096, 1024)
        bench("cfg4  256ch x 4096tap fmt6 B=1024 plain", 6, 256, 0, 4096, 1024, {"fir_impl": 0}, reps=3)
        bench("north 4096ch x 16bq+4096tap fmt6 B=1024", 6, 4096, 16, 4096, 1024, reps=5)
