#!/usr/bin/env python3
"""One-off sweep (GPU box): larger random chain shapes (up to 40 channels, 70 sections, 1500 taps, blocks up to
2500 frames, two blocks per run) through the parallel kernels against the oracle.  python tests/dev/gpu_chain_sweep.py LO HI
(tests/test_gpu_sweeps.py runs a slice of the seeds in the -m gpu suite)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from avdsp_amd import progbuilder as pb, runtime as rt
from oracle import pyoracle as po
from tests.fuzz_programs import stress_input


def _options():
    """AVDSP_SWEEP_OPTIONS="fir_impl=4,fir_lean=1,...": library options set on every runtime of the sweep"""
    return [(k, int(v)) for k, v in (kv.split("=") for kv in os.environ.get("AVDSP_SWEEP_OPTIONS", "").split(",") if kv)]


def run(lo, hi, formats=(2, 4, 6)):
    """seeds lo .. hi-1; returns (runs, list of mismatch descriptions)"""
    bad, n = [], 0
    opts = _options()
    for seed in range(lo, hi):
        rng = np.random.default_rng(seed + 7000)
        C = int(rng.choice([1, 2, 3, 5, 8, 17, 40])); S = int(rng.choice([0, 1, 2, 3, 7, 8, 9, 16, 17, 33, 70]))
        T = int(rng.choice([0, 1, 15, 16, 17, 100, 257, 900, 1500]))
        frames = int(rng.choice([1, 2, 15, 16, 17, 255, 256, 257, 1023, 1024, 1025, 2500]))
        for fmt in formats:
            taps = 0 if fmt == 2 else T
            if S == 0 and taps == 0: continue
            prog = pb.synth_program(fmt, C, S, taps, 5, 5, float(rng.choice([0.5, 1.0, 3.0])))
            o = po.OracleProgram(fmt, prog); r = rt.Runtime(fmt, prog)
            for k, v in opts: r.set_option(k, v)
            ok = True
            for blk in range(2):
                x = stress_input(rng, frames, C, fmt in (5, 6)) if rng.random() < 0.5 else pb.lcg_input(frames, C, fmt in (5, 6), seed=seed + blk)
                want = o.run_block(x, C, C, 0); got = r.run_block(x, C, C, 0)
                ok = ok and bool((got.view(np.uint32) == want.view(np.uint32)).all())
            ok = ok and bool((r.sync_state() == o.state).all())
            n += 1
            if not ok:
                bad.append(f"seed {seed} fmt {fmt} C {C} S {S} T {taps} frames {frames}")
            for k, v in opts: r.set_option(k, {'fir_impl': 1, 'fir_lean': -1, 'biquad_impl': 1}.get(k, 0))
            r.release()
    return n, bad


if __name__ == "__main__":
    n, bad = run(int(sys.argv[1]), int(sys.argv[2]), (2, 3, 4, 5, 6) if os.environ.get("AVDSP_SWEEP_ALL_FORMATS") else (2, 4, 6))
    for b in bad: print('MISMATCH', b)
    print('runs', n, 'bad', len(bad))
