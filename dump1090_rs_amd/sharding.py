"""Sharding of the demod_2400 path across GPUs: by buffer, no data-path collective.

The unit is the 131072-sample buffer (chunk): buffers are independent except for the ICAO
filter (reference src/utils.rs:44, src/lib.rs:36-44: a fresh zeroed MagnitudeBuffer per
call).  Each rank demodulates a contiguous range of buffers as its own stream, with its own
context/filter -- the same thing as running one dump1090_rs per SDR.  torch.distributed is
used for the barrier and for reducing the timing, nothing else.
"""
from __future__ import annotations

from typing import Tuple

CHUNK = 131072


def chunk_range(n_chunks: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [first, last) buffer range of `rank`; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(n_chunks, world)
    first = rank * base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


def sample_range(n_samples: int, world: int, rank: int) -> Tuple[int, int]:
    """[first, last) sample range of `rank` for a capture of n_samples (last buffer may be short)."""
    n_chunks = (n_samples + CHUNK - 1) // CHUNK
    a, b = chunk_range(n_chunks, world, rank)
    return min(a * CHUNK, n_samples), min(b * CHUNK, n_samples)


def reduce_timing(dist, elapsed_s: float, frames: int, device="cpu") -> Tuple[float, int]:
    """MAX of the elapsed time and SUM of the frame count over all ranks (the bench contract)."""
    import torch

    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    f = torch.tensor([frames], dtype=torch.int64, device=device)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    return float(t.item()), int(f.item())
