"""The N>1 path on CPU: world_size-2 (and 3) torch.distributed runs over gloo.  Each rank loads the UNSHARDED program
into the C-ABI library, calls dspRuntimeSetShard(rank, world) and asks dspRuntimeShardInfo (host-only lowering, no
GPU needed) which chains and IO windows are its own; it then processes that column slice of the same seeded input
block, and the ranks agree -- through an all_gather and an all_reduce, the only collectives the design uses, both at
the block boundary -- that the concatenated shards equal the unsharded result.  There is no GPU in this container,
so the arithmetic of a rank's slice is done by the ORACLE on the equivalent shard program (the checker standing in
for the kernels); what is under test is the library's cut, the slice bookkeeping and the collective calls bench.py
relies on.  The same test with the HIP kernels doing the work is tests/test_gpu_headline.py
(test_world_n_ranks_hip_path_over_gloo, -m gpu)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from avdsp_amd import progbuilder as pb
from avdsp_amd import sharding as sh


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fmt, C, S, T, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pyoracle as po
        from avdsp_amd import runtime as rt
        # the library's cut of the unsharded program (what the GPU path runs) ...
        r = rt.Runtime(fmt, pb.synth_program(fmt, C, S, T))
        r.set_shard(rank, world)
        info = r.shard_info()
        r.release()
        # ... must be the contiguous balanced range, with the IO windows of exactly those channels
        prog, lo, hi = sh.shard_program(fmt, C, S, T, world, rank)
        assert info == dict(total_chains=C, first_chain=lo, nchains=hi - lo, in_io_min=C + lo, in_io_max=C + hi - 1,
                            out_io_min=lo, out_io_max=hi - 1), info
        x = pb.lcg_input(B, C, fmt == 6, seed=31)
        xs = np.ascontiguousarray(x[:, info["in_io_min"] - C:info["in_io_max"] + 1 - C])
        assert (xs == sh.shard_block(x, world, rank)).all()
        o = po.OracleProgram(fmt, prog)
        y = o.run_block(xs, hi - lo, hi - lo)
        # block-boundary collectives: gather the (ragged) slices, reduce the checksums
        width = torch.tensor([hi - lo], dtype=torch.int64)
        widths = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(widths, width)
        wmax = int(max(w.item() for w in widths))
        pad = np.zeros((B, wmax), dtype=y.dtype); pad[:, :hi - lo] = y
        mine = torch.from_numpy(pad.view(np.int32).copy())
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        full = np.concatenate([p.numpy()[:, :int(w.item())] for p, w in zip(parts, widths)], axis=1)
        cks = torch.tensor([sh.block_checksum(y) % (1 << 62)], dtype=torch.int64)
        dist.all_reduce(cks, op=dist.ReduceOp.SUM)
        dist.barrier()
        if rank == 0:
            ref = po.OracleProgram(fmt, pb.synth_program(fmt, C, S, T)).run_block(x, C, C)
            q.put((bool((full.view(np.uint32) == ref.view(np.uint32)).all()),
                   int(cks.item()) == sum(sh.block_checksum(ref[:, a:b]) % (1 << 62)
                                          for a, b in (sh.shard_range(C, world, r) for r in range(world)))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,fmt,C,S,T", [(2, 6, 10, 3, 33), (2, 2, 7, 4, 0), (3, 6, 8, 2, 0)])
def test_sharded_equals_unsharded(world, fmt, C, S, T):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, fmt, C, S, T, 64, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    same, cks_ok = q.get(timeout=5)
    assert same and cks_ok


def test_shard_ranges_cover_and_balance():
    for C in (1, 7, 8, 4096, 16384, 16385):
        for world in (1, 2, 3, 4, 8):
            spans = [sh.shard_range(C, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == C
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
