"""Development probe: how dspRuntimeBlockAll arranges the shipped programs (cores, pieces, levels) and what it costs."""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from avdsp_amd import progbuilder as pb, runtime as rt
from oracle import pyoracle as po
for name, fmt, in_stride, in_base, out_stride in (("crossoverLV6.bin", 2, 8, 8, 8), ("dacdiy1.bin", 2, 8, 8, 8), ("tour_float.bin", 6, 16, 32, 32), ("dacfabriceo.bin", 2, 8, 8, 8), ("mydspcode.bin", 2, 8, 8, 8)):
    prog = np.fromfile("tests/golden/" + name, dtype=np.uint32)
    x = pb.lcg_input(4096, in_stride, fmt in (5, 6), seed=5)
    for split in (1, 0):
        o = po.OracleProgram(fmt, prog, fs=48000, random=1, dither=24)
        want = o.run_block(x, out_stride, in_base, 0, block=1024, frame=np.zeros(4096, dtype=np.uint32))
        r = rt.Runtime(fmt, prog, fs=48000, random=1, dither=24)
        r.set_option("strand_split", split)
        got = r.run_block_all(x, out_stride, in_base, 0, block=1024)
        ok = (got.view(np.uint32) == want.view(np.uint32)).all() and (r.sync_state() == o.state).all()
        t0 = time.perf_counter()
        for _ in range(5): r.run_block_all(x, out_stride, in_base, 0, block=4096)
        dt = (time.perf_counter() - t0) / 5 / 4096 * 1e6
        print(f"{name:18s} split {split}: cores {r.get_option('cores')} pieces {r.get_option('pieces')} levels {r.get_option('levels')}  match {ok}  wall {dt:.3f} us/frame (host buffers)", flush=True)
        r.set_option("strand_split", 1); r.release()
