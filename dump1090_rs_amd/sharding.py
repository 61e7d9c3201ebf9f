"""Sharding of the demod_2400 path across GPUs: by buffer, no data-path collective.

The unit is the 131072-sample buffer (chunk): buffers are independent except for the ICAO
filter (reference src/utils.rs:44, src/lib.rs:36-44: a fresh zeroed MagnitudeBuffer per
call).  Two ways to use N GPUs:

* independent streams (the bench, BASELINE config 4): each rank demodulates a contiguous
  range of buffers as its own stream, with its own context/filter -- the same thing as
  running one dump1090_rs per SDR.  torch.distributed is used for the barrier and for
  reducing the timing, nothing else.
* one capture, exact (`demod_sharded`): the ranks' results are merged into what a single
  stream would have produced.  The filter is the only coupling, so the exchange is tiny: the
  addresses each shard learned and the raw trial records (a few per buffer), each as two
  fixed-size tensor all-gathers (RCCL over xGMI, or gloo), then one ordered replay on rank 0.
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


def merge_records(shard_records, chunk_bases):
    """Concatenate the shards' raw trial records with `chunk` made global."""
    import numpy as np

    parts = []
    for rec, base in zip(shard_records, chunk_bases):
        r = rec.copy()
        r["chunk"] = r["chunk"] + np.uint32(base)
        parts.append(r)
    return np.concatenate(parts) if parts else np.zeros(0, dtype=shard_records[0].dtype)


def _coll_device(dist):
    """Where collective tensors live: the GPU for RCCL ("nccl"), the host for gloo."""
    import torch

    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def _all_gather_ragged(dist, mine_u8):
    """All ranks' byte strings (numpy uint8), gathered with two fixed-size tensor collectives: the
    lengths, then the payloads padded to the longest.  No pickling: this is the exchange step of the
    sharded form, and over RCCL it is one small all-gather over xGMI."""
    import numpy as np
    import torch

    dev = _coll_device(dist)
    world = dist.get_world_size()
    n = torch.tensor([mine_u8.size], dtype=torch.int64, device=dev)
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, n)
    counts = counts.cpu().numpy()
    width = max(int(counts.max()), 1)
    pad = np.zeros(width, dtype=np.uint8)
    pad[: mine_u8.size] = mine_u8
    out = torch.zeros(world * width, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(out, torch.from_numpy(pad).to(dev))
    out = out.cpu().numpy().reshape(world, width)
    return [out[r, : int(counts[r])] for r in range(world)]


def exchange_addresses(dist, mine):
    """Union of every rank's learned addresses (sorted u32): a few KB per rank."""
    import numpy as np

    mine = np.ascontiguousarray(np.asarray(mine, dtype=np.uint32))
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return mine
    parts = _all_gather_ragged(dist, mine.view(np.uint8))
    allv = np.concatenate([np.frombuffer(p.tobytes(), dtype=np.uint32) for p in parts]) if parts else np.zeros(0, np.uint32)
    return np.unique(allv)


def gather_records(dist, mine, chunk_base: int):
    """All shards' records on rank 0, `chunk` made global (None on the other ranks)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return merge_records([mine], [chunk_base])
    import numpy as np

    head = np.array([chunk_base], dtype=np.int64).view(np.uint8)
    body = np.ascontiguousarray(mine).view(np.uint8).reshape(-1)
    parts = _all_gather_ragged(dist, np.concatenate([head, body]))
    if dist.get_rank() != 0:
        return None
    recs = [np.frombuffer(p[8:].tobytes(), dtype=mine.dtype) for p in parts]
    bases = [int(np.frombuffer(p[:8].tobytes(), dtype=np.int64)[0]) for p in parts]
    return merge_records(recs, bases)


def demod_sharded(ctx, device_ptr: int, n_samples: int, chunk_base: int, dist=None, filter_table=None):
    """One capture cut into contiguous buffer ranges, one per rank; this rank's range is the
    `n_samples` at `device_ptr`, starting at global buffer index `chunk_base`.  Returns the
    single-stream frame list on rank 0 (None elsewhere): identical to demodulating the whole
    capture on one GPU or with the reference on a CPU (`filter_table`: rank 0's filter, read
    and updated; default empty = after icao_flush)."""
    from .context import replay_records

    import numpy as np

    learned = ctx.shard_scan(device_ptr, n_samples)
    if filter_table is not None:  # what the filter already holds can match from the first sample on
        learned = np.concatenate([learned, filter_table[filter_table != 0] & np.uint32(0xFFFFFF)])
    union = exchange_addresses(dist, learned)
    records = ctx.shard_finish(union)
    merged = gather_records(dist, records, chunk_base)
    if merged is None:
        return None
    return replay_records(merged, filter_table)
