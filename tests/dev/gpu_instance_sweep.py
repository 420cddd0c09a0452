#!/usr/bin/env python3
"""One-off sweep (GPU box): random chain programs in random numbers of INSTANCES (round 5: an instance is a further block of chains),
ragged blocks whose length -- and with it the distance between the instances' sample blocks -- changes from call to call, every
instance with an input of its own against the oracle: outputs of every block and each instance's data area at the end.
    python tests/dev/gpu_instance_sweep.py LO HI          (tests/test_gpu_sweeps.py runs a slice of the seeds in the -m gpu suite)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from avdsp_amd import progbuilder as pb, runtime as rt
from avdsp_amd import devmem as dm
from oracle import pyoracle as po
from tests.fuzz_programs import stress_input


def run(lo, hi, formats=(2, 3, 4, 5, 6)):
    import torch
    bad, n = [], 0
    for seed in range(lo, hi):
        rng = np.random.default_rng(seed + 91000)
        C = int(rng.choice([1, 2, 3, 5, 8, 12])); S = int(rng.choice([0, 1, 2, 3, 8, 9, 16, 17]))
        T = int(rng.choice([0, 0, 1, 16, 33, 257, 700]))
        ninst = int(rng.choice([1, 2, 3, 7, 16, 37]))
        blocks = [int(b) for b in rng.choice([1, 2, 16, 63, 64, 65, 255, 256, 257, 512, 513, 1024, 1500], size=int(rng.integers(2, 5)))]
        fmt = formats[seed % len(formats)]
        taps = 0 if fmt == 2 else T
        if S == 0 and taps == 0:
            S = 2
        prog = pb.synth_program(fmt, C, S, taps, 5, 5, float(rng.choice([0.5, 1.0, 3.0])))
        frames = sum(blocks)
        xs = np.stack([stress_input(rng, frames, C, fmt in (5, 6)) if rng.random() < 0.3 else pb.lcg_input(frames, C, fmt in (5, 6), seed=seed * 100 + i)
                       for i in range(ninst)])
        r = rt.Runtime(fmt, prog)
        if rng.random() < 0.4:
            r.set_option("overlap", 1)
        r.set_instances(ninst)
        got = np.zeros((ninst, frames, C), dtype=xs.dtype)
        st = torch.cuda.current_stream().cuda_stream
        pos = 0
        for b in blocks:
            xd = dm.to_device(np.ascontiguousarray(xs[:, pos:pos + b]))
            yd = torch.zeros((ninst, b, C), dtype=xd.dtype, device="cuda")
            torch.cuda.synchronize()
            r.run_block_all_instances_device(xd.data_ptr(), C, C, b * C, yd.data_ptr(), C, 0, b * C, b, st)
            torch.cuda.synchronize()
            got[:, pos:pos + b] = dm.to_host(yd)
            pos += b
        ok = True
        for i in range(ninst):
            o = po.OracleProgram(fmt, prog)
            want = np.concatenate([o.run_block(xs[i, p0:p0 + b], C, C) for p0, b in zip(np.cumsum([0] + blocks[:-1]), blocks)])
            ok = ok and bool((got[i].view(np.uint32) == want.view(np.uint32)).all()) and bool((r.instance_state(i) == o.state).all())
        n += 1
        if not ok:
            bad.append(f"seed {seed} fmt {fmt} C {C} S {S} T {taps} instances {ninst} blocks {blocks}")
        r.set_option("overlap", 0)
        r.release()
    return n, bad


if __name__ == "__main__":
    n, bad = run(int(sys.argv[1]), int(sys.argv[2]))
    for b in bad: print('MISMATCH', b)
    print('runs', n, 'bad', len(bad))
