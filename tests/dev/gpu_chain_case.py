"""One case of gpu_fuzz_sweep.py's chain part in detail: python tests/dev/gpu_chain_case.py SEED FMT  (which kernels disagree with the oracle, where)"""
import sys; sys.path.insert(0, '.')
import numpy as np
from avdsp_amd import progbuilder as pb, runtime as rt
from oracle import pyoracle as po
from tests.fuzz_programs import random_chain_case, stress_input

seed, fmt = int(sys.argv[1]), int(sys.argv[2])
rng, C, S, T, fmin, fmax, gain, fs, nf, dither = random_chain_case(seed + 100000)
taps = 0 if fmt == 2 else T
prog = pb.synth_program(2 if fmt == 2 else 6, C, S, taps, fmin, fmax, gain)
# the sweep draws the input and the block size from the same generator, in this order, once per format 2, 4, 6
for f in (2, 4, 6):
    tp = 0 if f == 2 else T
    if S == 0 and tp == 0:
        continue
    x = stress_input(rng, nf, C, f == 6); block = int(rng.choice([7, 64, nf]))
    if f == fmt:
        break
print(f"seed {seed} fmt {fmt}: C {C} S {S} taps {taps} fs {fs} frames {nf} block {block} gain {gain} dither {dither}")
print("  last input frames (hex):", [[f"{int(v):08x}" for v in row] for row in x.view(np.uint32)[-4:]])
o = po.OracleProgram(fmt, prog, fs=fs, dither=dither)
want = o.run_block(x, C, C, 0, block=block)
for bi, fi in ((1, 1), (2, 1), (0, 1), (1, 0), (0, 0)):
    r = rt.Runtime(fmt, prog, fs=fs, dither=dither)
    r.set_option("biquad_impl", bi); r.set_option("fir_impl", fi)
    got = r.run_block(x, C, C, 0, block=block)
    d = got.view(np.uint32) != want.view(np.uint32)
    st = r.sync_state() != o.state
    print(f"  biquad_impl {bi} fir_impl {fi}: {int(d.sum())} output words differ, {int(st.sum())} state words differ", end="")
    if d.any():
        fr, ch = np.nonzero(d)
        print(f"; first at frame {fr[0]} channel {ch[0]}: got {got.view(np.uint32)[fr[0], ch[0]]:08x} want {want.view(np.uint32)[fr[0], ch[0]]:08x}; input there {x.view(np.uint32)[max(fr[0]-3,0):fr[0]+1, ch[0]]}", end="")
    print()
    if st.any():
        idx = np.nonzero(st)[0]
        print("    state words", idx[:8].tolist(), "got", [f"{int(v) & 0xFFFFFFFF:08x}" for v in r.state[idx[:8]]], "want", [f"{int(v) & 0xFFFFFFFF:08x}" for v in o.state[idx[:8]]])
    r.set_option("biquad_impl", 1); r.set_option("fir_impl", 1)
    r.release()
