"""Channel sharding of a program across the GPUs of one node.

Every in-scope program is a set of per-channel chains with private state, so the
hot path shards by contiguous channel ranges with NO data-path collective: rank r owns channels
[r*C/N, (r+1)*C/N), keeps its slice of coefficients/taps/state resident, consumes its column slice
of each [B][C] input block and produces its slice of the output block.  RCCL (torch.distributed
"nccl") is used only for block-boundary barriers, the max-over-ranks timing and, on request, an
all-gather of the output slices / an all-reduce of their checksums."""
from __future__ import annotations

import numpy as np

from . import progbuilder as pb


def shard_range(total_channels: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous, balanced: the first (total % world) ranks get one channel more."""
    q, r = divmod(total_channels, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_program(fmt: int, total_channels: int, nsections: int, ntaps: int, world: int, rank: int,
                  **kw) -> tuple[np.ndarray, int, int]:
    """The rank's slice of the synthetic program of SURVEY.md 8(d): same filters and impulses as the
    global channel numbers, IO numbers local to the shard (inputs at C_local.., outputs at 0..)."""
    lo, hi = shard_range(total_channels, world, rank)
    return pb.synth_program(fmt, hi - lo, nsections, ntaps, channel_base=lo, **kw), lo, hi


def shard_block(x: np.ndarray, world: int, rank: int) -> np.ndarray:
    """Column slice of a frame-interleaved [B][C] block for this rank (a copy, contiguous)."""
    lo, hi = shard_range(x.shape[1], world, rank)
    return np.ascontiguousarray(x[:, lo:hi])


def block_checksum(y: np.ndarray) -> int:
    """Order-independent 64-bit checksum of a block's raw 32-bit words (sum mod 2^64), so that the
    sum of the shards' checksums equals the checksum of the concatenated block."""
    return int(np.ascontiguousarray(y).view(np.uint32).astype(np.uint64).sum(dtype=np.uint64))
