import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from avdsp_amd import runtime as rt
from tests.golden_recipes import GOLDEN_DIR
for name in ("crossoverLV6.bin", "dacdiy1.bin"):
    prog = np.fromfile(os.path.join(GOLDEN_DIR, name), dtype=np.uint32)
    r = rt.Runtime(2, prog, fs=48000, random=3, dither=24)
    s = np.zeros(64, dtype=np.int32)
    for _ in range(50):
        for k in range(len(r.cores)): r.run_frame(s, k)
    r.set_option("profile", 1); r.set_option("profile_stride", 1)
    n = 300
    t0 = time.perf_counter()
    for _ in range(n):
        for k in range(len(r.cores)): r.run_frame(s, k)
    wall = (time.perf_counter() - t0) / (n * len(r.cores)) * 1e6
    out = []
    for kind in (3, 5, 6):
        ms, launches = r.kernel_time(kind)
        if launches: out.append("kind %d: %.2f us x %d" % (kind, ms * 1e3 / launches, launches))
    print(name, "wall per core call (profiled) %.1f us;" % wall, "; ".join(out))
    r.set_option("profile", 0)
    r.release()
