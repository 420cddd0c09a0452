#!/usr/bin/env python3
"""bench.py -- Msamples/s (summed over all channels) of the AVDSP hot path on MI355X.

One step = one block of B frames pushed through every channel chain of the program
(LOAD_GAIN -> 16 biquads -> 4096-tap FIR -> SAT0DB -> STORE for the north-star workload) by
dspRuntimeBlockDevice(); inputs and outputs are resident in HBM before the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload north|cfg2|cfg3|cfg4|cfg5] [--block B]

N > 1 runs one rank per GPU, whoever starts it:
  * under a launcher (the driver's  python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...):
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment; WORLD_SIZE != --gpus is an ERROR (exit 2),
    not a warning -- a line that says n_gpus 1 for a job asked to run on 8 must not exist;
  * bare (python bench.py --gpus N, WORLD_SIZE unset): this process -- which has not touched the GPU and never will --
    times the CPU baseline, then starts that very launcher command as a CHILD process (N fresh rank processes; nothing
    is exec'ed over a process that has initialised HIP), hands the CPU figure to rank 0 through a temporary file
    (AVDSP_BENCH_CPU_JSON), lets rank 0's JSON line through on stdout and leaves with the children's exit code.
The experiment is the one BASELINE.json names: a FIXED program (north: 4096 channels;
cfg5: 16384) at 1, 2, 4 and 8 GPUs = STRONG scaling.  Every rank loads the same unsharded program and
calls dspRuntimeSetShard(rank, world): the library cuts the lowered chain list into contiguous balanced
ranges, the rank feeds its column slice of the [B][C] block and gets its slice of the output.  Channels
are independent, so there is NO data-path collective; torch.distributed (RCCL) is used for the barriers
and the max-over-ranks time.

Rank 0 prints ONE JSON line with, besides the contract fields:
  roofline     -- for the dominant kernel (the FIR on v_mfma_f64_16x16x4_f64 when the workload has
                  one, else the biquad cascade): algorithmic work per launch / average launch time
                  from the dispatches' own start / end stamps (hipExtLaunchKernel events on the launch
                  stream; roofline.timing says which launches were sampled).
  cpu_baseline -- the reference itself, compiled from /root/reference into oracle/_ref (kind
                  "reference"; one process per core this job may use, plus one thread alone), timed
                  on this box's host cores on a bounded sample of the same chain.  Only when those
                  binaries did not travel: the oracle, the CPU restatement (kind "port").
  verified / verified_ranks -- what every rank checked against the reference's pins before timing.
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
    # name: (format, channels IN TOTAL (all GPUs together), biquad sections, FIR taps, frames per block)
    "north": (6, 4096, 16, 4096, 1024),     # BASELINE.json north_star target
    "cfg2":  (6, 8, 8, 0, 256),
    "cfg3":  (6, 4096, 16, 0, 1024),
    "cfg3i": (2, 4096, 16, 0, 1024),        # same in int64 fixed point
    "cfg4":  (6, 256, 0, 4096, 1024),
    "cfg5":  (6, 16384, 8, 2048, 1024),     # BASELINE config 5: 16384 channels, 2048 per GPU at 8 GPUs
    "cfg5s": (6, 2048, 8, 2048, 1024),      # one 8-GPU shard's worth of cfg5 as a program of its own
    "north8": (6, 512, 16, 4096, 1024),     # what one rank of the north-star program runs at 8 GPUs
    # the cascade kernels with the chip FULL (tools/cascade_sweep.sh): cfg3 is exactly one wave per SIMD
    "cfg3x4":  (6, 16384, 16, 0, 1024), "cfg3x8":  (6, 32768, 16, 0, 1024),
    "cfg3ix4": (2, 16384, 16, 0, 1024), "cfg3ix8": (2, 32768, 16, 0, 1024),
}

# Peaks from /opt/skills/guides/MI355X_MICROARCH.md (HBM3E 8 TB/s) and the MI355X datasheet value
# quoted in SURVEY.md 8(d) / BASELINE.md 3 for FP64 (vector and matrix alike): 78.6 TFLOP/s.
PEAK_F64_TFLOPS = 78.6
PEAK_HBM_GBS = 8000.0
PEAK_F32_TFMAS = 157.3 / 2.0          # FP32 vector FMAs per second (157.3 TFLOP/s); 32x32-bit integer multiply-adds run at a quarter of it


def pmc_traffic(workload, kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same
    command (profiles/traffic.json, produced by tools/summarize_prof.py --traffic; FETCH_SIZE doubled
    per the gfx950 correction of MI355X_MICROARCH.md, WRITE_SIZE as read).  None when not profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f).get(workload, {}).get(kernel)
    except (OSError, ValueError):
        return None


def host_cores():
    """(cores this process may really use, os.cpu_count()): the affinity mask and the cgroup CPU quota both cap it;
    the box hands a one-GPU job a share of the host, not the whole machine."""
    n_all = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = n_all
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(-(-int(quota) // int(period)))))
    except (OSError, ValueError):
        pass
    return n, n_all


def _ref_run(fmt, S, T, B, nproc, budget_s):
    """`nproc` ref_driver processes, each with ONE channel of the same chain (the reference keeps one program per
    process in globals), steady state (FIR delay lines filled first: zero state is an early-out): (samples, seconds)"""
    import subprocess
    import tempfile
    from avdsp_amd import progbuilder as pb
    from oracle import pyoracle as po
    drv = os.path.join(po.REF_DIR, "ref_driver")
    lib = os.path.join(po.REF_DIR, f"libavdspref_{fmt}.so")
    warm = (T + B - 1) // B + 1
    with tempfile.TemporaryDirectory() as d:
        x = pb.lcg_input(B, 1, fmt in (5, 6))
        xin = os.path.join(d, "in.raw"); x.tofile(xin)
        progs = []
        for i in range(nproc):
            p = os.path.join(d, f"p{i}.bin")
            pb.synth_program(fmt, 1, S, T, channel_base=i).tofile(p)
            progs.append(p)

        def run(repeat):
            ps = [subprocess.Popen([drv, lib, str(fmt), progs[i], "0", "48000", "0", "31", xin,
                                    os.path.join(d, f"o{i}.raw"), str(B), str(B), "1", "1", "1", "0", "4",
                                    "-", str(repeat + warm), str(warm)], stdout=subprocess.PIPE, text=True)
                  for i in range(nproc)]
            outs = [p.communicate()[0] for p in ps]
            return max(float(o.split("elapsed=")[1].split()[0]) for o in outs)

        blocks = 4
        for _ in range(4):                                   # grow the sample until it fills about the budget
            dt = run(blocks)
            if dt >= 0.6 * budget_s:
                break
            blocks = max(blocks + 1, min(int(blocks * budget_s / max(dt, 1e-3)), 1 << 22))
    return nproc * blocks * B, dt, blocks


def _port_run(fmt, S, T, B, nthreads, budget_s):
    """the oracle (CPU restatement), one private program per thread (the C library releases the GIL)"""
    from avdsp_amd import progbuilder as pb
    from oracle import pyoracle as po
    po.build()
    x = pb.lcg_input(B, 1, fmt in (5, 6))
    warm_blocks = (T + B - 1) // B + 1
    progs = [po.OracleProgram(fmt, pb.synth_program(fmt, 1, S, T, channel_base=i)) for i in range(nthreads)]

    def many(fn):
        th = [threading.Thread(target=fn, args=(p,)) for p in progs]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        return time.perf_counter() - t0

    many(lambda p: [p.run_block(x, 1, 1) for _ in range(warm_blocks)])
    cal = max(16, min(B, 128))
    t0 = time.perf_counter()
    progs[0].run_block(x[:cal], 1, 1)
    per_sample = (time.perf_counter() - t0) / cal
    blocks = max(1, min(int(budget_s / max(per_sample * B, 1e-9)), 4096))
    dt = many(lambda p: [p.run_block(x, 1, 1) for _ in range(blocks)])
    return nthreads * blocks * B, dt, blocks


def cpu_baseline(fmt, S, T, B, budget_s=10.0):
    """the CPU runtime on THIS box's host cores, (i) one thread, (ii) every core this job may use, one
    channel of the same chain per core.  The compiled reference (oracle/_ref, kind "reference") when its binaries
    travelled with the snapshot, else the oracle (kind "port")."""
    from oracle import pyoracle as po
    cores, n_all = host_cores()
    kind, run = "port", _port_run
    if po.have_ref() and os.path.exists(os.path.join(po.REF_DIR, f"libavdspref_{fmt}.so")):
        kind, run = "reference", _ref_run
    try:
        s1, t1, b1 = run(fmt, S, T, B, 1, 0.35 * budget_s)
        sn, tn, bn = run(fmt, S, T, B, cores, budget_s)
    except Exception as e:                                   # fall through to the port, say why
        if kind == "port":
            raise
        print(f"cpu_baseline: reference run failed ({e}); timing the oracle instead", file=sys.stderr)
        kind = "port"
        s1, t1, b1 = _port_run(fmt, S, T, B, 1, 0.35 * budget_s)
        sn, tn, bn = _port_run(fmt, S, T, B, cores, budget_s)
    what = ("the compiled reference runtime (oracle/_ref, gcc -Ofast), one process per core" if kind == "reference"
            else "oracle/liboracle.so, one thread per core")
    return dict(value=sn / tn / 1e6, unit="Msamples/s", cores=cores, kind=kind,
                single_thread=s1 / t1 / 1e6, host_cpus=n_all,
                sample=f"{cores} x 1 ch x {bn} blocks of {B} frames of the same chain ({S} biquads + {T}-tap FIR, "
                       f"DSP_FORMAT {fmt}) through {what}, steady state, {tn:.1f} s; single_thread: 1 ch x {b1} blocks, {t1:.1f} s; "
                       f"cores = what this job may use (affinity mask and cgroup quota) of the host's {n_all} CPUs")


# BASELINE workloads whose every channel is pinned against the compiled reference (tests/golden/make_headline_goldens.py)
HEADLINE_CASE = {"north": "north_f6", "cfg3": "cfg3_f6", "cfg3i": "cfg3_f2", "cfg4": "cfg4_f6", "cfg5": "cfg5_f6"}


class VerificationError(Exception):
    pass


def verify_against_reference_pins(args, make_runtime, workload, shard_rank, shard_world, torch):
    """Before anything is timed: the benchmarked workload's HEADLINE input (seed 20260104, several blocks so that FIR histories
    fill) through a fresh runtime on the same dspRuntimeBlockDevice path with the same options, this rank's columns against the
    reference's pins -- per-channel sums of the output words of every block of the case, and (whole program on one rank) the
    SHA-256 of every block and of the final state.  Returns a description for the JSON line; raises VerificationError on a
    mismatch (main() lets every rank know before any of them leaves, so that an N > 1 job ends as one)."""
    import hashlib
    from avdsp_amd import devmem as dm
    from avdsp_amd import progbuilder as pb
    name = HEADLINE_CASE.get(workload)
    gdir = os.path.join(ROOT, "tests", "golden")
    if name is None or not os.path.exists(os.path.join(gdir, f"headline_{name}.npz")):
        return None
    with open(os.path.join(gdir, "headline_manifest.json")) as f:
        case = json.load(f)["cases"][name]
    g = np.load(os.path.join(gdir, f"headline_{name}.npz"))
    fmt, C, B, frames = case["fmt"], case["channels"], case["block"], case["frames"]
    x = pb.lcg_input(frames, C, fmt in (5, 6), seed=case["seed"])
    r = make_runtime()
    r.set_shard(shard_rank, shard_world)
    info = r.shard_info()
    Cl, in_base, out_base = info["nchains"], info["in_io_min"], info["out_io_min"]
    if Cl < 1:
        r.release()
        return f"{name}: shard {shard_rank}/{shard_world} holds no chains"
    xs = dm.to_device(x[:, in_base - C:in_base - C + Cl])
    ys = torch.zeros((frames, Cl), dtype=xs.dtype, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    # the pins are per 1024-frame block of the case; --block pushes the same frames through in calls of that many frames (the last
    # one ragged when it does not divide) -- "nframes successive dspRuntime_N calls" means the concatenation is the same bits
    Bcall = args.block if args.block > 0 else B
    for f0 in range(0, frames, Bcall):
        n = min(Bcall, frames - f0)
        r.run_block_device(xs[f0:].data_ptr(), Cl, in_base, ys[f0:].data_ptr(), Cl, out_base, n, stream)
    torch.cuda.synchronize()
    out = dm.to_host(ys)
    w = np.ascontiguousarray(out).view(np.uint32)
    col = w.sum(axis=0, dtype=np.uint64).astype(np.uint32)
    want = g["col_sum"][out_base:out_base + Cl]
    bad = np.nonzero(col != want)[0]
    if bad.size:
        raise VerificationError(f"shard {shard_rank}/{shard_world}: {bad.size} of {Cl} channels differ from the "
                                f"reference's pins of {name} (first: channel {out_base + int(bad[0])})")
    what = f"{name}: blocks 0..{frames // B - 1}, channels {out_base}..{out_base + Cl - 1}: per-channel word sums vs reference pins"
    if Bcall != B:
        what += f" (pushed through in calls of {Bcall} frames)"
    if shard_world == 1:
        for k, sha_want in enumerate(case["block_sha"]):
            if hashlib.sha256(np.ascontiguousarray(out[k * B:(k + 1) * B]).tobytes()).hexdigest() != sha_want:
                raise VerificationError(f"block {k} of {name} differs from the reference (SHA-256)")
        if hashlib.sha256(np.ascontiguousarray(r.sync_state()).tobytes()).hexdigest() != case["state_sha"]:
            raise VerificationError(f"final state of {name} differs from the reference (SHA-256)")
        what += " + SHA-256 of every block and of the final state"
    r.release()
    return what


def launch_ranks(args, fmt, S, T, B):
    """python bench.py --gpus N without a launcher: be the launcher.  This process makes no HIP call (counting devices does not
    initialise the GPU on this image); it times the CPU baseline, then runs
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <the same arguments>
    as a child -- N fresh rank processes -- with the CPU figure in a temporary file for rank 0.  stdout / stderr are the children's own
    (rank 0 prints the one JSON line); returns their exit code."""
    import socket
    import subprocess
    import tempfile
    backend = os.environ.get("AVDSP_DIST_BACKEND", "nccl")
    if backend == "nccl":
        import torch
        ndev = torch.cuda.device_count()
        if ndev < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but this node shows {ndev} GPU(s); one rank per GPU over RCCL "
                  f"(AVDSP_DIST_BACKEND=gloo rehearses several ranks on one card)", file=sys.stderr)
            return 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    with tempfile.TemporaryDirectory() as d:
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(fmt, S, T, B, budget_s=args.cpu_budget)
            path = os.path.join(d, "cpu_baseline.json")
            with open(path, "w") as f:
                json.dump(cpu, f)
            env["AVDSP_BENCH_CPU_JSON"] = path
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        sys.stdout.flush()
        return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="north", choices=sorted(WORKLOADS))
    ap.add_argument("--custom", default=None, help="FMT,CHANNELS,SECTIONS,TAPS,FRAMES: a synthetic chain program of any shape instead of a named workload (no reference pins: "
                    "nothing is verified, nothing is a BASELINE config -- tools/regime_scan.sh looks for launch arrangements that do not fit a shape)")
    ap.add_argument("--fir-impl", type=int, default=1, help="0 the reference's tap loop, 1 fir_tile (default), 2 fir_mfma (round 1), 3 fir_stream, 4 fir_flow (DESIGN.md 4.2)")
    ap.add_argument("--biquad-impl", type=int, default=1)
    ap.add_argument("--overlap", type=int, default=1, help="cascade of the next block under the FIR of this one (the blocks are resident in HBM, which is that mode's contract): 0 off, 1 on, 2 also the FIRs of consecutive blocks on two streams in turn")
    ap.add_argument("--ready-words", type=int, default=-1, help="under --overlap, how a block's FIR finds its cascades' block: 0 an event between the two queues, 1 per-chain ready words published by the cascade's waves, 2 ready words set by a kernel behind the cascade (no wait packet on the FIRs' stream); -1 (default) the library's choice by plan: 2 where the FIR is the bound, else 0")
    ap.add_argument("--fir-launch", type=int, default=-1, help="how the overlap mode enqueues its FIR launches: -1 the library's choice (default), 0 plain + recorded event, 1 the event on the dispatch's completion signal, 2 start and stop events on the dispatch")
    ap.add_argument("--ring-wait", type=int, default=1, help="under --overlap: who waits for the FIR three blocks back before a cascade reuses its ring positions: 1 the host (default), 0 the cascades' stream (a wait packet in front of every cascade)")
    ap.add_argument("--fir-lean", type=int, default=-1, help="fir_tile's lean chunk boundary: -1 the library's choice by plan (default), 0 never, 1 always")
    ap.add_argument("--fir-split", type=int, default=0, help="1: fir_tile launches of at most a tile per SIMD (cfg4) cut every tile's taps over two waves -- sums within 1e-6, NOT the reference's bits (so --no-verify's hash check is replaced by nothing: use for the A/B only)")
    ap.add_argument("--fir-rows", type=int, default=-1, help="fir_tile row tiles per wave: 0 auto, 1, 2, 4")
    ap.add_argument("--shard", default=None, help="RANK/WORLD: run that one shard of the program on this GPU alone (what one rank of a WORLD-GPU job does)")
    ap.add_argument("--host-buffers", action="store_true", help="also time dspRuntimeBlock_N with HOST buffers (PCIe inclusive), reported beside value")
    ap.add_argument("--host-split", type=int, default=-1, help="frames per piece of a host-pointer block (0 = whole block)")
    ap.add_argument("--host-pin", type=int, default=-1, help="pin the host buffers in place (bench.py keeps them allocated)")
    ap.add_argument("--settle", type=float, default=0.5, help="seconds of untimed steps in front of the warm-up: the chip raises its clock over the first ~0.1 s of load "
                    "(tools/fir_timeline.py: 2.15 GHz in-kernel after 6 blocks, 2.36 GHz after 200) and a short run would be timed on the ramp")
    ap.add_argument("--profile-stride", type=int, default=0, help="time every n-th launch of the dominant kernel in the timed region with its dispatch stamps (0: the default below)")
    ap.add_argument("--cu-split", type=int, default=0, help="experiment (DESIGN.md 5c): under --overlap the cascades' stream on that many CUs of its own (8, 16, 32 ...), the FIRs on a library stream on the others")
    ap.add_argument("--own-stream", action="store_true", help="hand the library a stream of this process's own (non-blocking) instead of the current (null) stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=10.0, help="seconds of CPU work for the all-cores leg of cpu_baseline (the single thread gets 0.35 of it)")
    ap.add_argument("--block", type=int, default=0, help="frames per block call (0: the workload's own, 1024 for the BASELINE configs): the host's period is the block "
                    "(linux/avdsp_plugin.c:71-98 hands dsp_transfer 64..1024 frames); blocks above 1024 frames are cut into 1024-frame launches by the library")
    ap.add_argument("--dist", action="store_true", help="initialise the process group and run every collective leg even at world 1 (under a launcher with "
                    "--nproc-per-node 1): what tests/ use to execute each RCCL call bench.py makes on a one-GPU box")
    ap.add_argument("--no-verify", action="store_true", help="skip the pre-timing check of this rank's columns against the reference's pins")
    ap.add_argument("--gather", action="store_true", help="(the default for N > 1) also time an RCCL all_gather of the ranks' [B][C/N] output blocks and an all_reduce of "
                    "their checksums (SURVEY.md 8e: block-boundary collectives), reported BESIDE value, never inside it")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the block-boundary collectives leg")
    args = ap.parse_args()

    if args.custom:
        WORKLOADS["custom"] = tuple(int(v) for v in args.custom.split(","))
        args.workload = "custom"
    fmt, C, S, T, B = WORKLOADS[args.workload]
    if args.block > 0:
        B = args.block
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    launched = "WORLD_SIZE" in os.environ
    if launched and int(os.environ["WORLD_SIZE"]) != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher's WORLD_SIZE is {os.environ['WORLD_SIZE']}: refusing to print a line "
              f"for a job of another size", file=sys.stderr)
        sys.exit(2)
    if not launched and args.gpus > 1:
        sys.exit(launch_ranks(args, fmt, S, T, B))

    import torch                      # first: its bundled HIP runtime must be the process's only one
    import torch.distributed as dist
    from avdsp_amd import devmem as dm
    from avdsp_amd import progbuilder as pb
    from avdsp_amd import runtime as rt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or (args.dist and launched)

    # The CPU leg runs FIRST, before this process touches the GPU: it starts one child process per host
    # core, and children must not be forked off a process that has initialised HIP.  At N > 1 it is rank 0's
    # too (the other ranks wait for it in init_process_group), unless the process that started the ranks has
    # already timed it (AVDSP_BENCH_CPU_JSON, see launch_ranks).
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        handed = os.environ.get("AVDSP_BENCH_CPU_JSON")
        if handed and os.path.exists(handed):
            with open(handed) as f:
                cpu = json.load(f)
        else:
            cpu = cpu_baseline(fmt, S, T, B, budget_s=args.cpu_budget)
    if not torch.cuda.is_available():
        sys.exit("bench.py: no GPU visible; the product path has no CPU fallback")
    # one rank per GPU on a real node; AVDSP_DIST_BACKEND=gloo lets several ranks rehearse the N > 1 path on
    # one card (RCCL refuses two ranks on the same device)
    backend = os.environ.get("AVDSP_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev:
        sys.exit(f"bench.py: {world} ranks over RCCL need {world} GPUs, this node shows {ndev} (AVDSP_DIST_BACKEND=gloo rehearses several ranks on one card)")
    device_index = local_rank % ndev
    torch.cuda.set_device(device_index)
    if use_dist:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend=backend)
    # STRONG scaling: the job is the fixed C-channel program; every rank loads it whole and runs its own chains
    shard_rank, shard_world = rank, world
    if args.shard:
        shard_rank, shard_world = (int(v) for v in args.shard.split("/"))
    prog_words = pb.synth_program(fmt, C, S, T)

    def make_runtime():
        rr = rt.Runtime(fmt, prog_words)
        if rr.rc < 0:
            sys.exit(f"dspRuntimeInit failed: {rr.rc} {rr.last_error()}")
        rr.set_option("device", device_index)
        rr.set_option("fir_impl", args.fir_impl)
        rr.set_option("biquad_impl", args.biquad_impl)
        rr.set_option("overlap", args.overlap)
        rr.set_option("ready_words", args.ready_words)
        rr.set_option("fir_split", args.fir_split)
        if args.fir_launch >= 0:
            rr.set_option("fir_launch", args.fir_launch)
        if args.fir_lean >= 0:
            rr.set_option("fir_lean", args.fir_lean)
        rr.set_option("ring_wait", args.ring_wait)
        if args.cu_split:
            rr.set_option("cu_split", args.cu_split)
        if args.fir_rows >= 0:
            rr.set_option("fir_rows", args.fir_rows)
        if args.host_split >= 0:
            rr.set_option("host_split", args.host_split)
        if args.host_pin >= 0:
            rr.set_option("host_pin", args.host_pin)
        return rr

    # What is about to be timed is checked first, on every rank: the same path, the same options, the reference's own bits
    # ... and the ranks of an N > 1 job leave TOGETHER when any of them fails: a rank that exited alone would leave the others in
    # their next barrier until the launcher's timeout
    verified, verify_error = None, None
    if not args.no_verify:
        try:
            verified = verify_against_reference_pins(args, make_runtime, args.workload, shard_rank, shard_world, torch)
            if os.environ.get("AVDSP_BENCH_FAIL_RANK") == str(rank):          # tests/: what the job does when one rank's check fails
                raise VerificationError("forced by AVDSP_BENCH_FAIL_RANK")
        except VerificationError as e:
            verify_error = str(e)
    verified_ranks = 1 if (verified is not None and not verify_error) else 0
    if use_dist:
        dev = "cuda" if backend == "nccl" else "cpu"
        ok = torch.tensor([0 if verify_error else 1, 1 if (verified is not None and not verify_error) else 0], dtype=torch.int64, device=dev)
        all_ok = ok.clone()
        dist.all_reduce(all_ok[:1], op=dist.ReduceOp.MIN)
        dist.all_reduce(ok[1:], op=dist.ReduceOp.SUM)
        verified_ranks = int(ok[1].item())
        if int(all_ok[0].item()) == 0:
            if verify_error:
                print(f"bench.py: VERIFICATION FAILED on rank {rank}: {verify_error}", file=sys.stderr, flush=True)
            dist.barrier()
            dist.destroy_process_group()
            sys.exit(3)
    elif verify_error:
        sys.exit(f"bench.py: VERIFICATION FAILED: {verify_error}")
    r = make_runtime()
    r.set_option("profile", 1)
    r.set_shard(shard_rank, shard_world)
    info = r.shard_info()
    Cl = info["nchains"]                                    # this rank's channels
    if Cl < 1:
        sys.exit(f"rank {rank}: no chains in shard {shard_rank}/{shard_world}")
    in_base, out_base = info["in_io_min"], info["out_io_min"]
    assert info["in_io_max"] - in_base + 1 == Cl and info["out_io_max"] - out_base + 1 == Cl

    xfull = pb.lcg_input(B, C, fmt == 6, seed=12345)
    xs = np.ascontiguousarray(xfull[:, in_base - C:in_base - C + Cl])      # the rank's column slice of the [B][C] block
    x = dm.to_device(xs)         # (through pinned memory: avdsp_amd/devmem.py)
    y = torch.zeros((B, Cl), dtype=x.dtype, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    if args.own_stream:
        torch.cuda.synchronize()
        own_stream = torch.cuda.Stream()
        stream = own_stream.cuda_stream

    # "overlap" 2 lets the FIRs of consecutive blocks run into each other: a block's output must then not be the buffer an earlier
    # call may still be writing (the mode's contract) -- three output blocks in turn
    ys = [y] + ([torch.zeros_like(y), torch.zeros_like(y)] if args.overlap >= 2 else [])
    turn = [0]

    def step():
        yo = ys[turn[0] % len(ys)]; turn[0] += 1
        r.run_block_device(x.data_ptr(), Cl, in_base, yo.data_ptr(), Cl, out_base, B, stream)

    # Every timed launch costs the stream an event pair (a few microseconds each).  In the timed region only the
    # kernel the roofline is quoted on carries one; the cascade in front of a FIR is timed over a few extra
    # untimed steps first (kernels_ms.biquad), back to back with the FIR (overlap off), which also gives the FIR
    # kernel's duration with the chip to itself (roofline.frac_alone).
    r.set_option("profile", 0)
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < args.settle:
        for _ in range(20):
            step()
        torch.cuda.synchronize()
    r.set_option("profile", 1)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    warm = [r.kernel_time(k) for k in (0, 1, 2)]            # (reading drops the warm-up launches from the kernel timers)
    # ... and says how long the dominant kernel's launch is: a launch that carries its stamps costs the stream ~5 us, 1 % of a 0.5-ms
    # FIR and 15 % of a 35-us cascade
    dom_ms = warm[1][0] / warm[1][1] if (T and warm[1][1]) else (warm[0][0] / warm[0][1] if warm[0][1] else 0.0)
    bq_side = fir_alone = None
    if T and S:
        r.set_option("overlap", 0)
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        bq_side = r.kernel_time(0)
        bq_pairs = r.get_option("timing_pairs_0")
        fir_alone = r.kernel_time(1)
        r.set_option("profile", 2 * (1 << 1))              # AVDSP_KERNEL_FIR only
        r.set_option("overlap", args.overlap)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        r.kernel_time(1)
    # every timed launch of a short run carries its event pair (>= 16 samples); longer runs sample every n-th (a pair costs the stream ~5 us)
    # (round 4: a launch that carries its stop event costs the stream ~5 us -- cfg3's step 39.8 us with every launch sampled, 35.8 with every
    # fourth, 34.7 with none -- so the timed region samples every fourth launch of the dominant kernel; roofline.timing says how many that made)
    if fir_alone is not None and fir_alone[1]:
        dom_ms = fir_alone[0] / fir_alone[1]
    # (round 5: EVERY launch when the dominant kernel's launch is >= 0.2 ms -- the driver's 20 steps then give 20 stamps, not 5)
    stride = 1 if dom_ms >= 0.2 else (4 if args.steps <= 64 else max(1, args.steps // 16))
    if args.profile_stride > 0:
        stride = args.profile_stride
    r.set_option("profile_stride", stride)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    rank_ms = [elapsed / args.steps * 1e3]
    if use_dist:
        dev = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        every = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(every, t)
        rank_ms = [float(v.item()) / args.steps * 1e3 for v in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # block-boundary collectives (opt-in, beside the metric): every rank's [B][C/N] output gathered on every rank, checksums summed
    gather = None
    if use_dist and not args.no_gather:
        # (shards differ by at most one chain: every rank hands over max(C/N) columns, the last one of a smaller shard unused)
        cmax = -(-C // world)
        ypad = y if Cl == cmax else torch.nn.functional.pad(y, (0, cmax - Cl))
        dev_y = ypad.contiguous() if backend == "nccl" else torch.from_numpy(dm.to_host(ypad))
        parts = [torch.empty_like(dev_y) for _ in range(world)]
        csum = dev_y.view(torch.int32).to(torch.int64).sum().reshape(1)
        for _ in range(3):
            dist.all_gather(parts, dev_y); dist.all_reduce(csum.clone())
        torch.cuda.synchronize(); dist.barrier()
        tg = time.perf_counter()
        n_g = 20
        for _ in range(n_g):
            dist.all_gather(parts, dev_y)
        torch.cuda.synchronize()
        t_gather = (time.perf_counter() - tg) / n_g
        tg = time.perf_counter()
        for _ in range(n_g):
            c2 = csum.clone(); dist.all_reduce(c2)
        torch.cuda.synchronize()
        t_reduce = (time.perf_counter() - tg) / n_g
        gather = {"all_gather_ms": t_gather * 1e3, "bytes_per_rank": int(dev_y.numel() * 4), "checksum_all_reduce_ms": t_reduce * 1e3,
                  "backend": backend}

    if bq_side is None:
        bq_ms, bq_n = r.kernel_time(0)
        bq_pairs = r.get_option("timing_pairs_0")
    else:
        bq_ms, bq_n = bq_side
    fir_ms, fir_n = r.kernel_time(1)
    fir_pairs = r.get_option("timing_pairs_1")
    # The timed kernels (fir_tile, biquad_row / biquad_pipe) carry their events in hipExtLaunchKernel's start / stop slots: the
    # dispatch's own start and end stamps, what rocprofv3's kernel trace reads, nothing on the stream -- no correction applies.
    # (Kernels launched the plain way are bracketed by two recorded events, which read ~5 us more than the kernel took:
    # kernels_ms.event_pair says how much on this box.)
    pairs = []
    for _ in range(60):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    pair_ms = float(np.median([a.elapsed_time(b) for a, b in pairs[10:]]))

    def timing_label(n, npairs):
        """how the library says it timed the launches it summed: by the dispatch's own stamps, or by an event pair around the launch"""
        if n and not npairs:
            return f"dispatch start/end stamps (hipExtLaunchKernel events) of {n} launches, every {stride}. launch of the timed region"
        if n and npairs == n:
            return f"event pairs recorded around {n} launches (each reads ~event_pair more than the kernel took)"
        return f"{n - npairs} launches by dispatch stamps, {npairs} by event pairs (those read ~event_pair more)" if n else "not timed"
    fir_raw, bq_raw = fir_ms / max(fir_n, 1), bq_ms / max(bq_n, 1)
    checksum = float(y.double().abs().sum().item()) if fmt == 6 else float(y.to(torch.float64).abs().sum().item())
    if not np.isfinite(checksum) or checksum == 0.0:
        sys.exit("bench.py: output block is empty or not finite")
    if r.get_option("ready_timeouts") != 0:
        sys.exit("bench.py: a FIR wave gave up waiting for its cascade's ready word -- the timed region is not valid")

    host_rate = None
    if args.host_buffers and rank == 0:
        # the reference-shaped boundary: dspRuntimeBlock_N with HOST pointers, PCIe crossings inside the timed region
        r.set_option("profile", 0)
        hy = np.zeros((B, Cl), dtype=xs.dtype)
        for _ in range(3):
            r.run_block(xs, Cl, in_base, out_base, out=hy)
        n = max(5, args.steps // 2)
        th = time.perf_counter()
        for _ in range(n):
            r.run_block(xs, Cl, in_base, out_base, out=hy)
        host_rate = Cl * B * n / (time.perf_counter() - th) / 1e6
        # ... and as a queue (dspRuntimeBlockSubmit / Wait): four blocks in flight, the host cycles through six buffer pairs
        ring = [(xs.copy(), np.zeros((B, Cl), dtype=xs.dtype)) for _ in range(6)]
        r.set_option("host_pin", 1)       # six buffer pairs, reused: keep their registrations (they outlive this leg)
        for k in range(30):
            r.submit_block(*ring[k % 6], in_base, out_base); r.wait_blocks(3)
        r.wait_blocks(0)
        nq = max(100, args.steps)
        th = time.perf_counter()
        for k in range(nq):
            r.submit_block(*ring[k % 6], in_base, out_base); r.wait_blocks(3)
        r.wait_blocks(0)
        host_queue_rate = Cl * B * nq / (time.perf_counter() - th) / 1e6
        if not np.isfinite(ring[(nq - 1) % 6][1].astype(np.float64)).all():
            sys.exit("bench.py: the queued host path left a non-finite output block")
        r.set_option("host_pin", 0)

    if rank == 0:
        ranks_share = shard_world if args.shard else world
        units = (Cl if args.shard else C) * B * args.steps   # samples all ranks processed (one shard alone: its own)
        value = units / elapsed / 1e6
        step_s = elapsed / args.steps
        # SURVEY.md 8(d) algorithmic bytes of one step of THIS rank: samples in+out, biquad state r+w and coefficients,
        # FIR history carry-in/out and taps
        step_bytes = 8.0 * Cl * B + 68.0 * S * Cl + (8.0 * (T - 1) * Cl + 4.0 * T * Cl if T else 0.0)
        # several ranks: rank 0's shard, if that shard was profiled alone (profiles/traffic.json has north and cfg5 in 8)
        traffic_key = f"{args.workload} shard {args.shard}" if args.shard else (args.workload if world == 1 else f"{args.workload} shard 0/{world}")
        fir_untimed = bool(T and not fir_n and fir_alone is not None and fir_alone[1])
        if fir_untimed:
            # --profile-stride beyond the run: no launch of the timed region carried its stamps (the un-sampled step); the roofline is
            # then the kernel alone, from the ten untimed steps in front of the timed region, and says so
            fir_raw, fir_n, fir_pairs = fir_alone[0] / fir_alone[1], fir_alone[1], 0
        if T and fir_n:
            # `frac` from the kernel's own start / end stamps of every timed launch (rocprofv3's kernel average of the same command
            # agrees, profiles/)
            per_launch = fir_raw * 1e-3
            # 2*T flop per sample; the Cl*B samples of a step are spread over fir_n/steps launches
            launches_per_step = (B + 1023) // 1024           # blocks longer than 1024 frames are cut into 1024-frame launches
            flops = 2.0 * T * B * Cl / launches_per_step
            ach = flops / per_launch / 1e12
            kname = {0: "fir_plain", 1: "fir_tile", 2: "fir_mfma", 3: "fir_stream", 4: "fir_flow"}[args.fir_impl]
            fir_bytes = (4.0 * Cl * B + 4.0 * (T - 1 + B) * Cl + 4.0 * T * Cl) / launches_per_step   # out, window, taps
            roof = dict(bound="mfma", kernel=kname, achieved=ach,
                        peak=PEAK_F64_TFLOPS, unit="TFLOP/s", frac=ach / PEAK_F64_TFLOPS,
                        timing=(timing_label(fir_n, fir_pairs) if not fir_untimed else
                                f"NO launch of the timed region was stamped (profile_stride {stride}): dispatch stamps of {fir_n} launches of the ten untimed steps in front of it, overlap off -- the kernel alone"),
                        traffic=pmc_traffic(traffic_key, kname), traffic_source="profiles/traffic.json (committed rocprofv3 --pmc passes of this command, not this run)",
                        hbm_frac=fir_bytes / per_launch / 1e9 / PEAK_HBM_GBS,
                        launch_ms=per_launch * 1e3, launches=fir_n)
            if fir_alone is not None and fir_alone[1]:
                # the same kernel without the next block's cascade beside it (10 untimed steps, overlap off)
                roof["launch_ms_alone"] = fir_alone[0] / fir_alone[1]
                roof["frac_alone"] = flops / (roof["launch_ms_alone"] * 1e-3) / 1e12 / PEAK_F64_TFLOPS
            # the FIR's flops of a step against the step's wall time: what the matrix pipe delivered over the whole timed region
            roof["frac_of_step"] = flops * launches_per_step / (elapsed / args.steps) / 1e12 / PEAK_F64_TFLOPS
            if args.overlap >= 2:
                roof["timing_note"] = ("overlap 2: the FIR launches of consecutive blocks are in flight together, a launch's own start-to-end "
                                       "time spans two kernels sharing the chip -- frac reads about half of what the pipe does; see frac_of_step")
        elif bq_n:
            # The cascade is a recurrence: what binds it is vector-instruction issue, not memory.  Both ceilings, the binding one first
            # (SURVEY.md 8d): double models 10 flop per section and sample against the FP64 vector peak; int64 5 32x32->64 MADs per
            # section and sample against a quarter of the FP32 FMA rate (v_mad_i64_i32); HBM: algorithmic bytes against 8 TB/s.
            per_launch = bq_raw * 1e-3
            lps = (B + 1023) // 1024                             # blocks longer than 1024 frames are cut into 1024-frame launches
            Bl = B / lps                                         # frames per launch
            nbytes = 8.0 * Cl * Bl + 48.0 * S * Cl + 20.0 * S * Cl   # samples in+out, state r+w, coefficients
            hbm_ach = nbytes / per_launch / 1e9
            kname = "biquad_simple" if not args.biquad_impl else ("biquad_pipe" if args.biquad_impl == 2 else "biquad_row_i64" if fmt == 2 else "biquad_row")
            if fmt == 2:
                ops, peak, unit = 5.0 * S * Cl * Bl, PEAK_F32_TFMAS / 4.0, "T MAD/s (v_mad_i64_i32; quarter of the FP32 FMA rate)"
            else:
                ops, peak, unit = 10.0 * S * Cl * Bl, PEAK_F64_TFLOPS, "TFLOP/s"
            ach = ops / per_launch / 1e12
            roof = dict(bound="valu", kernel=kname, achieved=ach, peak=peak, unit=unit, frac=ach / peak,
                        timing=timing_label(bq_n, bq_pairs),
                        hbm=dict(achieved=hbm_ach, peak=PEAK_HBM_GBS, unit="GB/s", frac=hbm_ach / PEAK_HBM_GBS),
                        hbm_frac=hbm_ach / PEAK_HBM_GBS,
                        traffic=pmc_traffic(traffic_key, kname) or pmc_traffic(traffic_key, "biquad_pipe"),
                        traffic_source="profiles/traffic.json (committed rocprofv3 --pmc passes of this command, not this run)",
                        launch_ms=per_launch * 1e3, launches=bq_n)
        else:
            roof = None
        shard_txt = (f"shard {shard_rank}/{shard_world} alone on one GPU ({Cl} ch)" if args.shard
                     else f"{C} ch in total, {Cl} ch/GPU by dspRuntimeSetShard(rank, {world})")
        line = {
            "metric": "Msamples/s (all ch) biquad+FIR chain", "value": value, "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64" if fmt != 2 else "int64", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {C} ch x ({S} biquads + {T}-tap FIR), block {B} frames, "
                                   f"DSP_FORMAT {fmt}; {shard_txt}, no data-path collective",
                       "channels": C, "channels_per_gpu": Cl, "sections": S, "taps": T, "block": B, "format": fmt,
                       "overlap": r.get_option("overlap"), "ready_words": r.get_option("ready_words"), "ready_mode": r.get_option("ready_mode"), "side_by_side": r.get_option("side_by_side"), "streams_remade": r.get_option("streams_remade"), "cu_split": r.get_option("cu_split"), "fir_split": r.get_option("fir_split"), "fir_launch": r.get_option("fir_launch"), "fir_lean": r.get_option("fir_lean"), "ring_wait": r.get_option("ring_wait"), "settle_s": args.settle, "profile_stride": stride},
            "roofline": roof,
            "hbm_frac_step": step_bytes / step_s / 1e9 / PEAK_HBM_GBS,
            "kernels_ms": {"biquad": bq_raw, "fir": fir_raw, "event_pair": pair_ms,
                           "biquad_timing": timing_label(bq_n, bq_pairs) + ("" if bq_side is None else "; ten untimed steps in front of the timed region, overlap off")},
        }
        line["verified"] = verified
        line["verified_ranks"] = verified_ranks          # ranks whose own columns matched the reference's pins (every rank checks; any mismatch ends the job)
        if use_dist:
            line["ranks"] = world
            line["rank_ms_per_step"] = rank_ms
            line["collectives"] = {"backend": backend, "world": world,
                                   "calls": "all_reduce MIN/SUM (verification), barrier, all_gather + all_reduce MAX (times)"
                                            + ("" if gather is None else ", all_gather of the [B][C/N] output blocks + all_reduce of their checksums")}
        if gather is not None:
            line["gather"] = gather
        if host_rate is not None:
            line["host_buffers_msamples_s"] = host_rate
            line["host_queue_msamples_s"] = host_queue_rate
        line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)
    r.set_option("profile_stride", 1)
    r.release()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
