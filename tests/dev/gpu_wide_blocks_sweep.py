#!/usr/bin/env python3
"""One-off sweep (GPU box): MANY chains (the automatic choice of 2 and 4 row tiles per FIR wave) x short and ragged blocks -- round 5's
regrouping of a workgroup's waves by the tiles a block has, at the chain counts where it matters; short taps so that the oracle keeps up.
    python tests/dev/gpu_wide_blocks_sweep.py LO HI"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from avdsp_amd import progbuilder as pb, runtime as rt
from oracle import pyoracle as po


def run(lo, hi):
    bad, n = [], 0
    for seed in range(lo, hi):
        rng = np.random.default_rng(seed + 55000)
        C = int(rng.choice([300, 512, 700, 1024, 1027, 2048, 2052, 3000])); S = int(rng.choice([0, 1, 2])); T = int(rng.choice([17, 64, 130]))
        fmt = int(rng.choice([4, 6]))
        blocks = [int(b) for b in rng.choice([1, 64, 128, 255, 256, 257, 384, 512, 513, 768, 769, 1024], size=int(rng.integers(2, 5)))]
        prog = pb.synth_program(fmt, C, S, T)
        frames = sum(blocks)
        x = pb.lcg_input(frames, C, fmt == 6, seed=seed)
        o = po.OracleProgram(fmt, prog); r = rt.Runtime(fmt, prog)
        if rng.random() < 0.5:
            r.set_option("fir_lean", int(rng.integers(0, 2)))
        ok, pos = True, 0
        for b in blocks:
            want = o.run_block(x[pos:pos + b], C, C); got = r.run_block(x[pos:pos + b], C, C)
            ok = ok and bool((got.view(np.uint32) == want.view(np.uint32)).all())
            pos += b
        ok = ok and bool((r.sync_state() == o.state).all())
        n += 1
        if not ok:
            bad.append(f"seed {seed} fmt {fmt} C {C} S {S} T {T} blocks {blocks}")
        r.set_option("fir_lean", -1)
        r.release()
    return n, bad


if __name__ == "__main__":
    n, bad = run(int(sys.argv[1]), int(sys.argv[2]))
    for b in bad: print('MISMATCH', b)
    print('runs', n, 'bad', len(bad))
