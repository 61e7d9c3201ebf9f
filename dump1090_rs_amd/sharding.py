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
  fixed-size all-gathers of HOST tensors over a gloo group (`host_group`) -- the exchange goes
  through the host, no RCCL collective touches the data path (north_star) -- then
  one ordered replay on rank 0.

Also here: which host cores a rank should run on (`plan_affinity`): the cores of its GPU's NUMA
node, shared out among the ranks whose GPUs hang off the same node.
"""
from __future__ import annotations

import glob
import os
from typing import Dict, List, Optional, Sequence, Tuple

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


_HOST_GROUPS: Dict[int, object] = {}


def host_group(dist):
    """The process group the sharded form exchanges over: host tensors over gloo, whatever the
    default backend is.  (With RCCL as the default backend a gloo group over the same ranks is
    created once -- a collective call: every rank makes it, the first demod_sharded does.)"""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return None
    if dist.get_backend() == "gloo":
        return dist.group.WORLD
    key = id(dist.group.WORLD)
    if key not in _HOST_GROUPS:
        _HOST_GROUPS[key] = dist.new_group(backend="gloo")
    return _HOST_GROUPS[key]


def _all_gather_ragged(dist, mine_u8, group=None):
    """All ranks' byte strings (numpy uint8), gathered with two fixed-size collectives on host
    tensors: the lengths, then the payloads padded to the longest.  No pickling."""
    import numpy as np
    import torch

    group = group if group is not None else host_group(dist)
    world = dist.get_world_size()
    n = torch.tensor([mine_u8.size], dtype=torch.int64)
    counts = torch.zeros(world, dtype=torch.int64)
    dist.all_gather_into_tensor(counts, n, group=group)
    counts = counts.numpy()
    width = max(int(counts.max()), 1)
    pad = np.zeros(width, dtype=np.uint8)
    pad[: mine_u8.size] = mine_u8
    out = torch.zeros(world * width, dtype=torch.uint8)
    dist.all_gather_into_tensor(out, torch.from_numpy(pad), group=group)
    out = out.numpy().reshape(world, width)
    return [out[r, : int(counts[r])] for r in range(world)]


def exchange_addresses(dist, mine, group=None):
    """Union of every rank's learned addresses (sorted u32): a few KB per rank."""
    import numpy as np

    mine = np.ascontiguousarray(np.asarray(mine, dtype=np.uint32))
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return mine
    parts = _all_gather_ragged(dist, mine.view(np.uint8), group)
    allv = np.concatenate([np.frombuffer(p.tobytes(), dtype=np.uint32) for p in parts]) if parts else np.zeros(0, np.uint32)
    return np.unique(allv)


def gather_records(dist, mine, chunk_base: int, group=None):
    """All shards' records on rank 0, `chunk` made global (None on the other ranks)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return merge_records([mine], [chunk_base])
    import numpy as np

    head = np.array([chunk_base], dtype=np.int64).view(np.uint8)
    body = np.ascontiguousarray(mine).view(np.uint8).reshape(-1)
    parts = _all_gather_ragged(dist, np.concatenate([head, body]), group)
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


class ShardPipeline:
    """demod_sharded over a sequence of captures with the two phases of consecutive steps overlapped.

    Step i is scan -> address exchange -> finish (match + records) -> record gather -> replay on rank 0.
    The second half of step i (a worker thread, context i % 2, its own gloo group) runs while the main
    thread scans step i + 1 on the other context and exchanges its addresses over the first group: two
    groups, each used by one thread only, so every rank issues each group's collectives in the same order.
    Every step starts from a flushed filter (the bench's step; `icao_flush` per call as
    benches/demod_benchmark.rs:9 does), so the steps are independent and their results come back in
    submission order from `submit` (the step two back) and `drain`.
    """

    def __init__(self, contexts, dist=None):
        from concurrent.futures import ThreadPoolExecutor

        assert len(contexts) == 2
        self.ctx = list(contexts)
        self.dist = dist if (dist is not None and dist.is_initialized() and dist.get_world_size() > 1) else None
        self.g_addr = self.g_rec = None
        if self.dist is not None:   # (collective calls: every rank constructs its pipeline at the same point)
            self.g_addr = self.dist.new_group(backend="gloo")
            self.g_rec = self.dist.new_group(backend="gloo")
        self.pool = ThreadPoolExecutor(max_workers=1)
        self.pending = [None, None]
        self.step = 0

    def _finish(self, ctx, union, chunk_base):
        from .context import replay_records

        records = ctx.shard_finish(union)
        merged = gather_records(self.dist, records, chunk_base, self.g_rec)
        return None if merged is None else replay_records(merged)

    def submit(self, device_ptr: int, n_samples: int, chunk_base: int):
        """Start step `self.step`; returns the result of the step that last used this step's context
        (two steps back: the frame list on rank 0, None on the other ranks), or None for the first two."""
        k = self.step % 2
        done = self.pending[k].result() if self.pending[k] is not None else None
        ctx = self.ctx[k]
        ctx.icao_flush()
        learned = ctx.shard_scan(device_ptr, n_samples)
        union = exchange_addresses(self.dist, learned, self.g_addr)
        self.pending[k] = self.pool.submit(self._finish, ctx, union, chunk_base)
        self.step += 1
        return done

    def drain(self):
        """The results still in flight, oldest first."""
        order = [self.step % 2, (self.step + 1) % 2]
        out = []
        for k in order:
            if self.pending[k] is not None:
                out.append(self.pending[k].result())
                self.pending[k] = None
        return out

    def close(self):
        self.drain()
        self.pool.shutdown()


# ------------------------------------------------------------------------------------------------
# host cores for a rank: those of its GPU's NUMA node
# ------------------------------------------------------------------------------------------------
def _read(path: str) -> Optional[str]:
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def parse_cpulist(text: str) -> List[int]:
    """"0-3,8,10-11" -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)."""
    cpus: List[int] = []
    for part in (text or "").split(","):
        part = part.strip()
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.extend(range(int(a), int(b or a) + 1))
    return cpus


def gpu_numa_nodes(sysfs: str = "/sys") -> List[Tuple[str, int]]:
    """(PCI address, NUMA node) of every AMD GPU function in PCI order -- the order HIP numbers the
    devices in, unless a *_VISIBLE_DEVICES variable picks or reorders them (`visible_devices`)."""
    seen = {}
    for dev in glob.glob(os.path.join(sysfs, "class/drm/card*/device")):
        if _read(os.path.join(dev, "vendor")) != "0x1002":
            continue
        cls = _read(os.path.join(dev, "class")) or ""
        if not (cls.startswith("0x03") or cls.startswith("0x12")):  # display controller / processing accelerator
            continue
        addr = os.path.basename(os.path.realpath(dev))
        node = _read(os.path.join(dev, "numa_node"))
        seen[addr] = int(node) if node not in (None, "") else -1
    return sorted(seen.items())


def visible_devices(env=os.environ) -> Optional[List[int]]:
    """The physical indices HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES
    select (None: all devices, in order; unparsable entries such as UUIDs: None as well)."""
    for name in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = env.get(name)
        if v:
            try:
                return [int(x) for x in v.split(",") if x.strip() != ""]
            except ValueError:
                return None
    return None


def plan_affinity(local_rank: int, local_world: int, gpus: Sequence[Tuple[str, int]],
                  node_cpus: Dict[int, Sequence[int]], allowed: Sequence[int],
                  visible: Optional[Sequence[int]] = None) -> dict:
    """The cores rank `local_rank` of `local_world` on this host should run on.

    Its GPU is device `local_rank` of the visible ones; it gets the allowed cores of that GPU's NUMA
    node, and when several ranks' GPUs hang off one node the node's cores are cut into equal
    contiguous parts, one per rank (a rank spends most of a step in HIP calls on one thread: what
    matters is that eight ranks do not pile onto the same cores or run across the socket link).
    Unknown topology (no sysfs entry, node -1): the allowed cores cut into local_world parts."""
    allowed = sorted(set(allowed))
    order = list(visible) if visible else list(range(len(gpus)))

    def device_of(r: int) -> int:
        """the visible device rank r runs on: more ranks than GPUs (an oversubscribed gloo run) wrap
        around, as bench.py's Env does (local_rank % device_count)"""
        return r % len(order) if order else -1

    def node_of(r: int) -> int:
        d = device_of(r)
        if d >= 0 and 0 <= order[d] < len(gpus):
            return gpus[order[d]][1]
        return -1

    node = node_of(local_rank)
    cpus = [c for c in node_cpus.get(node, ()) if c in set(allowed)] if node >= 0 else []
    if cpus:
        # every rank whose (effective) GPU hangs off this node shares the node's cores -- ranks that
        # share a GPU included
        sharers = [r for r in range(local_world) if node_of(r) == node]
        k, m, source = sharers.index(local_rank), len(sharers), "numa node of the rank's GPU"
    else:
        cpus, k, m, source = allowed, local_rank, max(1, local_world), "no NUMA information: even split of the allowed cores"
    per = max(1, len(cpus) // m)
    mine = cpus[k * per:(k + 1) * per] if k < m - 1 else cpus[k * per:]
    if not mine:
        mine = cpus
    d = device_of(local_rank)
    return {"numa_node": node, "cpus": mine, "ranks_on_node": m, "source": source,
            "gpu": gpus[order[d]][0] if d >= 0 and 0 <= order[d] < len(gpus) else None}


def pin_to_gpu_numa_node(local_rank: int, local_world: int, sysfs: str = "/sys") -> dict:
    """sched_setaffinity of this process to plan_affinity()'s cores.  Reads sysfs only: safe to call
    before anything has touched the GPU (bench.py does, first thing in a rank)."""
    gpus = gpu_numa_nodes(sysfs)
    node_cpus = {}
    for d in glob.glob(os.path.join(sysfs, "devices/system/node/node[0-9]*")):
        node_cpus[int(os.path.basename(d)[4:])] = parse_cpulist(_read(os.path.join(d, "cpulist")) or "")
    allowed = sorted(os.sched_getaffinity(0))
    plan = plan_affinity(local_rank, local_world, gpus, node_cpus, allowed, visible_devices())
    try:
        os.sched_setaffinity(0, plan["cpus"])
        plan["applied"] = True
    except OSError as e:  # a container may forbid it: report, do not fail the bench
        plan["applied"] = False
        plan["error"] = str(e)
    plan["cpus_before"] = len(allowed)
    return plan
