#!/usr/bin/env python3
"""Development probe (GPU box): python tests/dev/gpu_fuzz_one.py SEED FMT -- where does the interpreter differ from the oracle?"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from avdsp_amd import progbuilder as pb, runtime as rt
from oracle import pyoracle as po
from tests.fuzz_programs import IN_BASE, N_IN, N_OUT, random_program
seed, fmt = int(sys.argv[1]), int(sys.argv[2])
L = rt.lib(); names = [(C.c_char_p * 62).in_dll(L, 'dspOpcodeText')[i].decode().strip() for i in range(62)]
prog = random_program(seed, fmt)
fs, block = [48000, 48000, 96000][seed % 3], [1, 64, 300][seed % 3]
x = pb.lcg_input(300, N_IN, fmt in (5, 6), seed=seed)
o = po.OracleProgram(fmt, prog, fs=fs, random=seed, dither=24); r = rt.Runtime(fmt, prog, fs=fs, random=seed, dither=24)
want = o.run_block(x, N_OUT, IN_BASE, 0, scratch_len=48, block=block); got = r.run_block(x, N_OUT, IN_BASE, 0, block=block)
g, w = got.view(np.uint32), want.view(np.uint32)
cols = np.nonzero((g != w).any(axis=0))[0]
for c in cols:
    fr = np.nonzero(g[:, c] != w[:, c])[0]
    print('col', c, 'bad frames', len(fr), 'first', fr[:4], 'gpu', [hex(v) for v in g[fr[:3], c]], 'oracle', [hex(v) for v in w[fr[:3], c]])
r.sync_state(); n = int(prog[1]) + int(prog[2])
sd = np.nonzero(r.buf[12:n] != o.buf[12:n])[0] + 12
print('buffer diffs at data offsets', sd - int(prog[1]), [(hex(r.buf[i]), hex(o.buf[i])) for i in sd[:6]])
i = 0; lines = []
while i < prog[1]:
    op = int(prog[i] >> 16); sk = int(prog[i] & 0xffff)
    if op == 3: lines.append('--CORE')
    elif op not in (1, 4, 5): lines.append(f'{i} {names[op]} {[int(np.int32(v)) for v in prog[i+1:i+min(sk,7)]]}')
    if sk == 0: break
    i += sk
for c in cols:
    idx = [k for k, l in enumerate(lines) if f'DSP_STORE [{c}]' in l]
    for j in idx: print('\n'.join(lines[max(0, j - 14):j + 1])); print('....')
