#!/usr/bin/env python3
"""bench.py -- Msamples/s (summed over all channels) of the AVDSP hot path on MI355X.

One step = one block of B frames pushed through every channel chain of the program
(LOAD_GAIN -> 16 biquads -> 4096-tap FIR -> SAT0DB -> STORE for the north-star workload) by
dspRuntimeBlockDevice(); inputs and outputs are resident in HBM before the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload north|cfg2|cfg3|cfg4|cfg5]

N > 1 is launched by the driver as  python -m torch.distributed.run --nproc-per-node N bench.py ...
one rank per GPU.  Channels are independent, so ranks shard channels with NO data-path collective
(weak scaling: every rank runs the workload's channel count with its own slice of the global
channel numbering); torch.distributed (RCCL) is used for the barriers and the max-over-ranks time.

Rank 0 prints ONE JSON line with, besides the contract fields:
  roofline     -- for the dominant kernel (the FIR on v_mfma_f64_16x16x4_f64 when the workload has
                  one, else the biquad cascade): algorithmic work per launch / average launch time
                  measured with HIP events recorded by the library on the launch stream.
  cpu_baseline -- the oracle (CPU restatement of the reference, bit-identical to it on the golden
                  vectors) timed on this box's host cores on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (format, channels per GPU, biquad sections, FIR taps, frames per block)
    "north": (6, 4096, 16, 4096, 1024),     # BASELINE.json north_star target
    "cfg2":  (6, 8, 8, 0, 256),
    "cfg3":  (6, 4096, 16, 0, 1024),
    "cfg3i": (2, 4096, 16, 0, 1024),        # same in int64 fixed point
    "cfg4":  (6, 256, 0, 4096, 1024),
    "cfg5":  (6, 2048, 8, 2048, 1024),      # 16384 channels over 8 GPUs
}

# Peaks from /opt/skills/guides/MI355X_MICROARCH.md (HBM3E 8 TB/s) and the MI355X datasheet value
# quoted in SURVEY.md 8(d) / BASELINE.md 3 for FP64 (vector and matrix alike): 78.6 TFLOP/s.
PEAK_F64_TFLOPS = 78.6
PEAK_HBM_GBS = 8000.0


def pmc_traffic(workload, kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same
    command (profiles/traffic.json, produced by tools/summarize_prof.py --traffic; FETCH_SIZE doubled
    per the gfx950 correction of MI355X_MICROARCH.md, WRITE_SIZE as read).  None when not profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f).get(workload, {}).get(kernel)
    except (OSError, ValueError):
        return None


def cpu_baseline_reference(fmt, S, T, B, budget_s=12.0):
    """The COMPILED REFERENCE (oracle/_ref, built in the build container from /root/reference with its
    own -Ofast flags; binaries travel to the GPU box): one ref_driver process per host core, each with
    one channel of the same chain (the reference keeps one program per process in globals)."""
    import subprocess
    import tempfile
    from avdsp_amd import progbuilder as pb
    from oracle import pyoracle as po
    cores = max(1, min(os.cpu_count() or 1, 64))
    drv = os.path.join(po.REF_DIR, "ref_driver")
    lib = os.path.join(po.REF_DIR, f"libavdspref_{fmt}.so")
    warm = (T + B - 1) // B + 1                              # passes that fill the FIR delay line (zero state is an early-out)
    with tempfile.TemporaryDirectory() as d:
        x = pb.lcg_input(B, 1, fmt in (5, 6))
        xin = os.path.join(d, "in.raw"); x.tofile(xin)
        progs = []
        for i in range(cores):
            p = os.path.join(d, f"p{i}.bin")
            pb.synth_program(fmt, 1, S, T, channel_base=i).tofile(p)
            progs.append(p)

        def run(repeat):
            ps = [subprocess.Popen([drv, lib, str(fmt), progs[i], "0", "48000", "0", "31", xin,
                                    os.path.join(d, f"o{i}.raw"), str(B), str(B), "1", "1", "1", "0", "4",
                                    "-", str(repeat + warm), str(warm)], stdout=subprocess.PIPE, text=True)
                  for i in range(cores)]
            outs = [p.communicate()[0] for p in ps]
            el = [float(o.split("elapsed=")[1].split()[0]) for o in outs]
            return max(el)

        blocks = 4
        for _ in range(4):                                   # grow the sample until it fills about the budget
            dt = run(blocks)
            if dt >= 0.6 * budget_s:
                break
            blocks = max(blocks + 1, min(int(blocks * budget_s / max(dt, 1e-3)), 1 << 22))
    samples = cores * blocks * B
    return dict(value=samples / dt / 1e6, unit="Msamples/s", cores=cores, kind="reference",
                sample=f"{cores} processes x 1 ch x {blocks} blocks of {B} frames of the same chain ({S} biquads + "
                       f"{T}-tap FIR, DSP_FORMAT {fmt}) through the compiled reference runtime (oracle/_ref, gcc -Ofast), "
                       f"steady state, {dt:.1f} s")


def cpu_baseline(fmt, S, T, B, budget_s=12.0):
    """Reference binaries when they travelled with the snapshot, else the oracle (CPU restatement):
    one private program per thread, channels partitioned across threads."""
    from avdsp_amd import progbuilder as pb
    from oracle import pyoracle as po
    if po.have_ref() and os.path.exists(os.path.join(po.REF_DIR, f"libavdspref_{fmt}.so")):
        try:
            return cpu_baseline_reference(fmt, S, T, B, budget_s)
        except Exception as e:                               # fall through to the port, say why
            print(f"cpu_baseline: reference run failed ({e}); timing the oracle instead", file=sys.stderr)
    po.build()
    cores = max(1, min(os.cpu_count() or 1, 64))
    # size the sample from a short calibration so the whole leg stays near budget_s
    ch_per_thread = 1
    prog = pb.synth_program(fmt, ch_per_thread, S, T)
    x = pb.lcg_input(B, ch_per_thread, fmt in (5, 6))
    o = po.OracleProgram(fmt, prog)
    warm_blocks = (T + B - 1) // B + 1                      # fill the FIR delay line: zero state takes an early-out
    cal_frames = max(16, min(B, 128))
    for _ in range(warm_blocks):
        o.run_block(x, ch_per_thread, ch_per_thread)
    t0 = time.perf_counter()
    o.run_block(x[:cal_frames], ch_per_thread, ch_per_thread)
    per_sample = (time.perf_counter() - t0) / (cal_frames * ch_per_thread)
    blocks = int(budget_s / max(per_sample * B * ch_per_thread, 1e-9))
    blocks = max(1, min(blocks, 4096))
    progs = [po.OracleProgram(fmt, pb.synth_program(fmt, ch_per_thread, S, T, channel_base=i)) for i in range(cores)]

    def warm(p):
        for _ in range(warm_blocks):
            p.run_block(x, ch_per_thread, ch_per_thread)

    th = [threading.Thread(target=warm, args=(p,)) for p in progs]
    for t in th:
        t.start()
    for t in th:
        t.join()

    def work(p):
        for _ in range(blocks):
            p.run_block(x, ch_per_thread, ch_per_thread)

    th = [threading.Thread(target=work, args=(p,)) for p in progs]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    samples = cores * blocks * B * ch_per_thread
    return dict(value=samples / dt / 1e6, unit="Msamples/s", cores=cores, kind="port",
                sample=f"{cores} threads x {ch_per_thread} ch x {blocks} blocks of {B} frames of the same chain "
                       f"({S} biquads + {T}-tap FIR, DSP_FORMAT {fmt}), oracle/liboracle.so, {dt:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="north", choices=sorted(WORKLOADS))
    ap.add_argument("--fir-impl", type=int, default=1)
    ap.add_argument("--biquad-impl", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch                      # first: its bundled HIP runtime must be the process's only one
    import torch.distributed as dist
    from avdsp_amd import progbuilder as pb
    from avdsp_amd import runtime as rt
    from avdsp_amd import sharding as sh

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    fmt, C, S, T, B = WORKLOADS[args.workload]
    # The CPU leg runs FIRST, before this process touches the GPU: it starts one child process per host
    # core, and children must not be forked off a process that has initialised HIP.
    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(fmt, S, T, B)
    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible; the product path has no CPU fallback")
    # one rank per GPU on a real node; AVDSP_DIST_BACKEND=gloo lets several ranks rehearse the N > 1 path on
    # one card (RCCL refuses two ranks on the same device)
    backend = os.environ.get("AVDSP_DIST_BACKEND", "nccl")
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend=backend)
    # weak scaling: the job is C*world channels, this rank owns the contiguous slice [rank*C, (rank+1)*C)
    prog, ch_lo, ch_hi = sh.shard_program(fmt, C * world, S, T, world, rank)
    assert ch_hi - ch_lo == C
    r = rt.Runtime(fmt, prog)
    if r.rc < 0:
        sys.exit(f"dspRuntimeInit failed: {r.rc} {r.last_error()}")
    r.set_option("device", device_index)
    r.set_option("fir_impl", args.fir_impl)
    r.set_option("biquad_impl", args.biquad_impl)
    r.set_option("profile", 1)

    x = torch.from_numpy(pb.lcg_input(B, C, fmt == 6, seed=12345 + rank)).cuda()
    y = torch.zeros((B, C), dtype=x.dtype, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        r.run_block_device(x.data_ptr(), C, C, y.data_ptr(), C, 0, B, stream)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    for k in (0, 1, 2):
        r.kernel_time(k)                                   # drop warm-up launches from the kernel timers
    # Every timed launch costs the stream an event pair (a few microseconds each).  In the timed region only the
    # kernel the roofline is quoted on carries one; the cascade in front of a FIR is timed over a few extra
    # untimed steps first (kernels_ms.biquad).
    bq_side = None
    if T and S:
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        bq_side = r.kernel_time(0)
        r.kernel_time(1)
        r.set_option("profile", 2 * (1 << 1))              # AVDSP_KERNEL_FIR only
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    bq_ms, bq_n = r.kernel_time(0) if bq_side is None else bq_side
    fir_ms, fir_n = r.kernel_time(1)
    checksum = float(y.double().abs().sum().item()) if fmt == 6 else float(y.to(torch.float64).abs().sum().item())
    if not np.isfinite(checksum) or checksum == 0.0:
        sys.exit("bench.py: output block is empty or not finite")

    if rank == 0:
        total_samples = world * C * B * args.steps
        value = total_samples / elapsed / 1e6
        if T and fir_n:
            per_launch = fir_ms / fir_n * 1e-3
            # 2*T flop per sample; the C*B samples of a step are spread over fir_n/steps launches
            flops = 2.0 * T * B * C * args.steps / fir_n
            ach = flops / per_launch / 1e12
            kname = "fir_mfma" if args.fir_impl else "fir_plain"
            roof = dict(bound="mfma", kernel=kname, achieved=ach,
                        peak=PEAK_F64_TFLOPS, unit="TFLOP/s", frac=ach / PEAK_F64_TFLOPS, traffic=pmc_traffic(args.workload, kname),
                        launch_ms=per_launch * 1e3, launches=fir_n)
        elif bq_n:
            per_launch = bq_ms / bq_n * 1e-3
            nbytes = 8.0 * C * B + 48.0 * S * C + 20.0 * S * C   # samples in+out, state r+w, coefficients
            ach = nbytes / per_launch / 1e9
            kname = "biquad_pipe" if args.biquad_impl else "biquad_simple"
            roof = dict(bound="hbm", kernel=kname, achieved=ach,
                        peak=PEAK_HBM_GBS, unit="GB/s", frac=ach / PEAK_HBM_GBS, traffic=pmc_traffic(args.workload, kname),
                        launch_ms=per_launch * 1e3, launches=bq_n)
        else:
            roof = None
        line = {
            "metric": "Msamples/s (all ch) biquad+FIR chain", "value": value, "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64" if fmt != 2 else "int64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {C} ch/GPU x ({S} biquads + {T}-tap FIR), block {B} frames, "
                                   f"DSP_FORMAT {fmt}, channels sharded {world}-way, no data-path collective",
                       "channels_per_gpu": C, "sections": S, "taps": T, "block": B, "format": fmt},
            "roofline": roof,
            "kernels_ms": {"biquad": bq_ms / max(bq_n, 1), "fir": fir_ms / max(fir_n, 1)},
        }
        line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)
    r.release()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
