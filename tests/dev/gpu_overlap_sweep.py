#!/usr/bin/env python3
"""One-off sweep (GPU box): random cascade + FIR chain programs through the overlap mode (the cascades of later blocks under the FIRs of
earlier ones; device-resident blocks enqueued back to back without synchronisation) in its arrangements -- "ready_words" 0 / 1 / 2,
"ring_wait" 0 / 1, "overlap" 1 / 2, fir_tile's two boundaries, fir_flow -- against the oracle: outputs of every block and the final state.
    python tests/dev/gpu_overlap_sweep.py LO HI"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from avdsp_amd import progbuilder as pb, runtime as rt
from avdsp_amd import devmem as dm
from oracle import pyoracle as po
from tests.fuzz_programs import stress_input

ARR = [dict(overlap=1, ready_words=0, ring_wait=1), dict(overlap=1, ready_words=2, ring_wait=1), dict(overlap=1, ready_words=2, ring_wait=0),
       dict(overlap=1, ready_words=1, ring_wait=1), dict(overlap=2, ready_words=0, ring_wait=1), dict(overlap=1, ready_words=2, ring_wait=1, fir_lean=1),
       dict(overlap=1, ready_words=0, ring_wait=0, fir_lean=0), dict(overlap=1, ready_words=2, ring_wait=1, fir_impl=4), dict(overlap=1, ready_words=-1, ring_wait=1)]


def run(lo, hi):
    bad, n = [], 0
    for seed in range(lo, hi):
        rng = np.random.default_rng(seed + 9000)
        fmt = int(rng.choice([4, 6]))
        C = int(rng.choice([1, 3, 8, 17, 40, 130])); S = int(rng.choice([1, 2, 7, 16, 17, 33])); T = int(rng.choice([1, 16, 100, 257, 900, 2100]))
        B = int(rng.choice([64, 256, 300, 1024])); nb = int(rng.choice([4, 7, 9]))
        prog = pb.synth_program(fmt, C, S, T, 5, 5, float(rng.choice([0.5, 1.0, 3.0])))
        x = stress_input(rng, B * nb, C, fmt == 6) if rng.random() < 0.3 else pb.lcg_input(B * nb, C, fmt == 6, seed=seed)
        o = po.OracleProgram(fmt, prog)
        want = o.run_block(x, C, C, block=B)
        arr = ARR[seed % len(ARR)]
        r = rt.Runtime(fmt, prog)
        for k, v in arr.items(): r.set_option(k, v)
        xd = [dm.to_device(np.ascontiguousarray(x[k * B:(k + 1) * B])) for k in range(nb)]
        yd = [torch.zeros((B, C), dtype=xd[0].dtype, device="cuda") for _ in range(nb)]
        torch.cuda.synchronize()
        st = torch.cuda.current_stream().cuda_stream
        for k in range(nb):
            r.run_block_device(xd[k].data_ptr(), C, C, yd[k].data_ptr(), C, 0, B, st)
        torch.cuda.synchronize()
        got = np.concatenate([dm.to_host(y) for y in yd])
        ok = bool((got.view(np.uint32) == want.view(np.uint32)).all()) and bool((r.sync_state() == o.state).all()) and r.get_option("ready_timeouts") == 0
        n += 1
        if not ok:
            bad.append(f"seed {seed} fmt {fmt} C {C} S {S} T {T} B {B} x {nb} {arr}")
        for k in arr: r.set_option(k, {"ready_words": -1, "ring_wait": 1, "fir_lean": -1, "fir_impl": 1}.get(k, 0))
        r.release()
    return n, bad


if __name__ == "__main__":
    n, bad = run(int(sys.argv[1]), int(sys.argv[2]))
    for b in bad: print("MISMATCH", b)
    print("runs", n, "bad", len(bad))
