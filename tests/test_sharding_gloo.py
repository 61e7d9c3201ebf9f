"""The N > 1 path on CPU: world_size 2 over gloo.  Ranks take contiguous buffer ranges of one
capture, demodulate them as independent streams (the oracle stands in for the GPU, which
this container does not have), and reduce timing/frames the way bench.py does."""
import os
import socket
import sys

import numpy as np
import pytest

from dump1090_rs_amd import sharding, synth
from tests.conftest import ROOT


def test_chunk_range_partitions_exactly():
    for n in (0, 1, 7, 8, 512, 4097):
        for world in (1, 2, 3, 8):
            spans = [sharding.chunk_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert sharding.sample_range(5 * 131072 + 100, 2, 1) == (3 * 131072, 5 * 131072 + 100)
    with pytest.raises(ValueError):
        sharding.chunk_range(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_samples, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from oracle import binding
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        iq = synth.make_iq(n_samples, n_bursts=40, seed=77)
        a, b = sharding.sample_range(n_samples, world, rank)
        dist.barrier()
        msgs, _ = binding.Oracle().demod_iq(iq[a:b])          # independent stream, own filter
        elapsed, frames = sharding.reduce_timing(dist, 0.25 * (rank + 1), len(msgs))
        dist.barrier()
        q.put((rank, (a, b), [(m["chunk"], m["j"], m["buffer"]) for m in msgs], elapsed, frames))
    finally:
        dist.destroy_process_group()


def test_two_ranks_independent_streams_and_reduction(oracle_mod):
    import torch.multiprocessing as mp
    world, n = 2, 5 * 131072 + 4321
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    iq = synth.make_iq(n, n_bursts=40, seed=77)
    total = 0
    for rank, (a, b), frames, elapsed, nframes in got:
        assert (a, b) == sharding.sample_range(n, world, rank)
        want, _ = oracle_mod.Oracle().demod_iq(iq[a:b])
        assert frames == [(m["chunk"], m["j"], m["buffer"]) for m in want]
        total += len(want)
        assert elapsed == 0.5            # MAX over ranks of 0.25, 0.5
    assert all(g[4] == total for g in got) and total >= 35
    # buffers are independent (no carry-over), so per-shard (chunk, j, frame) lists, re-based,
    # are exactly the single-stream list whenever the filter does not matter: DF17 of a
    # fresh address always decodes (score 1400 or 1800)
    single, _ = oracle_mod.Oracle().demod_iq(iq)
    merged = [(c + sharding.chunk_range(6, world, r)[0], j, f) for r, _, fr, _, _ in got for c, j, f in fr]
    assert [(m["chunk"], m["j"], m["buffer"]) for m in single if m["buffer"][0] >> 3 == 17] == \
        [x for x in merged if x[2][0] >> 3 == 17]
