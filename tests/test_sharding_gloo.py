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


# --- one capture, exact merge (sharding.demod_sharded's host side) -------------------------
def _coupled_capture(n_samples):
    """Bursts whose result depends on the filter state across the shard boundary: an
    address/parity DF4 (mode_s/mod.rs:56-72) for an address that a DF17 in buffer 1 teaches,
    sent before that (must be dropped), later in the first shard and in the second shard."""
    iq = synth.make_iq(n_samples, n_bursts=120, seed=78, n_icao=6, df11_every=4)
    icao = 0x4840D6
    body = bytes([0x20, 0x00, 0x05, 0x30])
    df4 = body + (synth.crc24(body) ^ icao).to_bytes(3, "big")
    at = lambda chunk, j: 5 * (chunk * 131072 + j)
    synth.add_bursts(iq, [synth.Burst(at(0, 20000), 22000, 1, df4),
                          synth.Burst(at(1, 60000), 22000, 2, synth.df17_frame(icao, 77)),
                          synth.Burst(at(2, 90000), 22000, 5, df4),
                          synth.Burst(at(4, 5000), 22000, 9, df4),
                          synth.Burst(at(5, 70000), 22000, 3, df4)])
    return iq, df4


def _shard_trials(oracle_mod, iq):
    """Stand-in for adsb_shard_scan/finish on a box without a GPU: every trial the oracle
    slices in this shard (a superset of what the device keeps; the replay decides)."""
    import ctypes as C
    from dump1090_rs_amd.context import TRIAL_DTYPE
    L = oracle_mod.lib()
    parts = []
    for c, off in enumerate(range(0, len(iq), 131072)):
        mb = oracle_mod.OrcMagBuf()
        part = np.ascontiguousarray(iq[off:off + 131072])
        L.orc_to_mag(part.ctypes.data, len(part), C.byref(mb))
        buf = np.zeros(5 * 131072 // 8, dtype=TRIAL_DTYPE)
        n = L.orc_all_trials(C.byref(mb), c, buf.ctypes.data, len(buf))
        assert n <= len(buf)
        parts.append(buf[:n].copy())
    return np.concatenate(parts) if parts else np.zeros(0, dtype=TRIAL_DTYPE)


def _merge_worker(rank, world, port, n_samples, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from oracle import binding
    from dump1090_rs_amd.context import replay_records
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        iq, _ = _coupled_capture(n_samples)
        a, b = sharding.sample_range(n_samples, world, rank)
        recs = _shard_trials(binding, iq[a:b])
        # the address exchange: every rank ends up with the same union
        mine = np.unique(np.array([int.from_bytes(bytes(r["msg"][1:4]), "big") for r in recs
                                   if r["msg"][0] >> 3 == 17][:50], dtype=np.uint32))
        union = sharding.exchange_addresses(dist, mine)
        merged = sharding.gather_records(dist, recs, a // 131072)
        out = None
        if rank == 0:
            msgs = replay_records(merged)
            out = [(m.chunk, m.j, m.try_phase, m.score, m.msg, m.signal_level) for m in msgs]
        q.put((rank, union.tolist(), mine.tolist(), out))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_exact_merge_equals_single_stream(oracle_mod, hip_lib):
    """Shards gathered on rank 0 and replayed once in global order == the whole capture on one
    stream, including address/parity frames whose address was learned in the other shard."""
    import torch.multiprocessing as mp
    world, n = 2, 6 * 131072 + 999
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_merge_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][1] == got[1][1] == sorted(set(got[0][2]) | set(got[1][2]))
    iq, df4 = _coupled_capture(n)
    single, _ = oracle_mod.Oracle().demod_iq(iq)
    want = [(m["chunk"], m["j"], m["try_phase"], m["score"], m["msg"], m["signal_level"]) for m in single]
    assert got[0][3] == want and got[1][3] is None
    # the capture really exercises the coupling: the DF4 decodes in buffers 2, 4 and 5 (never
    # in buffer 0, before its address is known); independent streams would lose 4 and 5
    assert sorted(m["chunk"] for m in single if m["buffer"] == df4) == [2, 4, 5]
    per_shard = []
    for r in range(world):
        a, b = sharding.sample_range(n, world, r)
        per_shard += oracle_mod.Oracle().demod_iq(iq[a:b])[0]
    assert len([m for m in per_shard if m["buffer"] == df4]) == 1


def test_bench_starts_its_own_ranks_from_a_plain_invocation():
    """`python bench.py --gpus 2` with no launcher: the parent starts two ranks under
    torch.distributed.run before touching any device, they rendezvous on 127.0.0.1, and rank 0's
    single JSON line comes out of the parent (--dry-run: no GPU here, no measurement)."""
    import json
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--dry-run"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rank_sum"] == 3 and out["dry_run"] is True


def test_bench_parent_process_never_imports_torch_before_launching():
    """The launching parent must not initialise a GPU runtime (a process that has may not start
    another program): it does not import torch at all."""
    import subprocess
    code = ("import sys, bench; sys.argv=['bench.py','--gpus','2','--dry-run'];\n"
            "import subprocess as sp\n"
            "real = sp.run\n"
            "def fake(cmd, **kw):\n"
            "    assert 'torch' not in sys.modules, 'torch imported before the ranks were started'\n"
            "    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node=2' in cmd\n"
            "    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'\n"
            "    class R: returncode = 0; stdout = '{\"metric\": \"x\"}\\n'\n"
            "    return R()\n"
            "sp.run = fake\n"
            "try:\n"
            "    bench.main()\n"
            "except SystemExit as e:\n"
            "    assert e.code == 0\n"
            "print('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=str(ROOT), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr


# --- world size 4, uneven ranges; the exchange over the host group; core affinity -----------------
def _ws4_worker(rank, world, port, n_chunks, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from dump1090_rs_amd.context import TRIAL_DTYPE
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        a, b = sharding.chunk_range(n_chunks, world, rank)
        # a rank's "learned addresses" and "trial records": sized by its (uneven) range, rank 3's empty
        mine = np.arange(a, b, dtype=np.uint32) * 7 + 1 if rank != 3 else np.zeros(0, np.uint32)
        recs = np.zeros(3 * (b - a) if rank != 3 else 0, dtype=TRIAL_DTYPE)
        recs["chunk"] = np.repeat(np.arange(b - a, dtype=np.uint32), 3)[: len(recs)]
        recs["j_tp"] = (rank + 1) * 1000 + np.arange(len(recs), dtype=np.uint32)
        group = sharding.host_group(dist)
        union = sharding.exchange_addresses(dist, mine)
        merged = sharding.gather_records(dist, recs, a)
        # the same over two private groups, from two threads at once (what ShardPipeline does)
        import threading
        g1, g2 = dist.new_group(backend="gloo"), dist.new_group(backend="gloo")
        box = {}
        th = threading.Thread(target=lambda: box.__setitem__("m", sharding.gather_records(dist, recs, a, g2)))
        th.start()
        u2 = sharding.exchange_addresses(dist, mine, g1)
        th.join()
        elapsed, frames = sharding.reduce_timing(dist, 0.1 * (rank + 1), b - a)
        q.put((rank, (a, b), union.tolist(), None if merged is None else
               [(int(r["chunk"]), int(r["j_tp"])) for r in merged], u2.tolist() == union.tolist(),
               (box["m"] is None) == (merged is None) and (merged is None or bool((box["m"] == merged).all())),
               group is not None, elapsed, frames))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_four_ranks_uneven_ranges_exchange_over_the_host_group():
    """World size 4 over 10 buffers (3, 3, 2, 2), one rank with nothing to contribute: the address union
    and the record gather (fixed-size host-tensor all-gathers over a gloo group) give every rank the
    same union and rank 0 all records re-based to global buffer numbers, also when the two exchanges run
    from two threads over two groups as ShardPipeline issues them."""
    import torch.multiprocessing as mp
    world, n_chunks = 4, 10
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ws4_worker, args=(r, world, port, n_chunks, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [g[1] for g in got] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    want_union = sorted(int(c) * 7 + 1 for c in range(0, 8))          # rank 3 contributed none
    assert all(g[2] == want_union for g in got)
    assert all(g[3] is None for g in got[1:])
    merged = got[0][3]
    assert len(merged) == 3 * 8
    # re-based: rank 1's local buffers 0..2 are global 3..5, rank 2's 0..1 are 6..7
    assert [c for c, _ in merged] == [0] * 3 + [1] * 3 + [2] * 3 + [3] * 3 + [4] * 3 + [5] * 3 + [6] * 3 + [7] * 3
    assert [jt // 1000 for _, jt in merged] == [1] * 9 + [2] * 9 + [3] * 6
    assert all(g[4] and g[5] and g[6] for g in got)
    assert all(abs(g[7] - 0.4) < 1e-12 and g[8] == n_chunks for g in got)


def test_rank_affinity_plan_follows_the_gpus_numa_nodes(tmp_path):
    """plan_affinity / gpu_numa_nodes on a fake sysfs: eight GPUs on two NUMA nodes, ranks get disjoint
    quarters of their node's cores; HIP_VISIBLE_DEVICES reorders; unknown topology falls back to an even split."""
    sysfs = tmp_path / "sys"
    for i in range(8):
        d = sysfs / "devices" / "pci0000:00" / f"0000:{0x10 + 0x10 * i:02x}:00.0"
        d.mkdir(parents=True)
        (d / "vendor").write_text("0x1002\n")
        (d / "class").write_text("0x120000\n")
        (d / "numa_node").write_text(f"{0 if i < 4 else 1}\n")
        card = sysfs / "class" / "drm" / f"card{7 - i}"          # card numbers do not follow PCI order
        card.mkdir(parents=True)
        (card / "device").symlink_to(d)
    other = sysfs / "devices" / "pci0000:00" / "0000:05:00.0"     # another vendor's display controller
    other.mkdir(parents=True)
    (other / "vendor").write_text("0x1a03\n")
    (other / "class").write_text("0x030000\n")
    (other / "numa_node").write_text("0\n")
    c8 = sysfs / "class" / "drm" / "card8"
    c8.mkdir(parents=True)
    (c8 / "device").symlink_to(other)
    gpus = sharding.gpu_numa_nodes(str(sysfs))
    assert [n for _, n in gpus] == [0, 0, 0, 0, 1, 1, 1, 1] and gpus[0][0] == "0000:10:00.0"
    node_cpus = {0: list(range(0, 64)) + list(range(128, 192)), 1: list(range(64, 128)) + list(range(192, 256))}
    plans = [sharding.plan_affinity(r, 8, gpus, node_cpus, range(256)) for r in range(8)]
    assert [p["numa_node"] for p in plans] == [0] * 4 + [1] * 4
    assert all(len(p["cpus"]) == 32 for p in plans)
    assert set().union(*[set(p["cpus"]) for p in plans[:4]]) == set(node_cpus[0])
    assert not set(plans[0]["cpus"]) & set(plans[1]["cpus"]) and not set(plans[4]["cpus"]) & set(node_cpus[0])
    # a restricted starting affinity is respected; a reordering *_VISIBLE_DEVICES maps rank -> physical GPU
    p = sharding.plan_affinity(0, 2, gpus, node_cpus, range(0, 16), visible=[5, 1])
    assert p["numa_node"] == 1 and p["source"].startswith("no NUMA") and len(p["cpus"]) == 8   # node 1 has none of 0..15
    p = sharding.plan_affinity(1, 2, gpus, node_cpus, range(0, 16), visible=[5, 1])
    assert p["numa_node"] == 0 and p["cpus"] == list(range(0, 16)) and p["gpu"] == "0000:20:00.0"
    assert sharding.visible_devices({"HIP_VISIBLE_DEVICES": "3,1"}) == [3, 1]
    assert sharding.visible_devices({"ROCR_VISIBLE_DEVICES": "GPU-abc"}) is None and sharding.visible_devices({}) is None
    # more ranks than GPUs (eight gloo ranks on a one-GPU box): rank r runs on device r % 1, and the eight
    # share that GPU's node cores in disjoint parts instead of overlapping rank 0's
    one = gpus[:1]
    plans = [sharding.plan_affinity(r, 8, one, node_cpus, range(256)) for r in range(8)]
    assert all(p["numa_node"] == 0 and p["ranks_on_node"] == 8 and p["gpu"] == "0000:10:00.0" for p in plans)
    assert all(p["source"].startswith("numa node") and len(p["cpus"]) == 16 for p in plans)
    assert sum(len(p["cpus"]) for p in plans) == len(set().union(*[set(p["cpus"]) for p in plans])) == 128
    # no sysfs at all: even split of what the process may run on
    p = sharding.plan_affinity(2, 4, [], {}, range(8))
    assert p["cpus"] == [4, 5] and p["numa_node"] == -1
    assert sharding.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and sharding.parse_cpulist("") == []


def _one_process_leg_worker(rank, world, port, q):
    """bench.py's `one_process_n_devices_leg` with the GPU work stubbed out: what is under test is the choreography --
    every rank creates the host-side (gloo) group, rank 0 alone works (here: sleeps, or raises), the others wait at the
    group's barrier and nobody is left behind."""
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import contextlib
    import time
    import types
    import torch
    import torch.distributed as dist
    import bench
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        env = types.SimpleNamespace(torch=torch, dist=dist, rank=rank, world=world, local_rank=0,
                                    all_cores=lambda: contextlib.nullcontext())
        args = types.SimpleNamespace(steps=3, capture_chunks=8)
        torch.cuda.device_count = lambda: 1     # (rank 0 sees one device: shards wrap around onto it)
        seen = {}

        def fake_leg(e, a, device_sets, steps):
            seen["device_sets"] = device_sets
            time.sleep(1.5)                     # the other rank must still be waiting when this returns
            return {"runs": [], "parity_checked": True}

        bench.config4_leg = fake_leg
        t0 = time.perf_counter()
        leg = bench.one_process_n_devices_leg(env, args)
        waited = time.perf_counter() - t0
        # ... and when the leg cannot run, the line still comes out and the waiting ranks are released
        def broken_leg(e, a, device_sets, steps):
            raise MemoryError("no room for the capture")
        bench.config4_leg = broken_leg
        leg2 = bench.one_process_n_devices_leg(env, args)
        q.put((rank, leg, waited, seen.get("device_sets"), leg2))
    finally:
        dist.destroy_process_group()


def test_one_process_leg_behind_the_timed_region_releases_every_rank():
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_one_process_leg_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, leg0, waited0, sets0, leg0b), (r1, leg1, waited1, sets1, leg1b) = got
    assert (r0, r1) == (0, 1)
    assert leg0["parity_checked"] is True and leg0["devices_visible_to_rank0"] == 1 and sets0 == [[0, 0]] and leg1 is None and sets1 is None
    assert waited0 >= 1.4 and waited1 >= 1.2          # rank 1 sat at the host-side barrier while rank 0 worked
    assert leg0b["parity_checked"] is None and "MemoryError" in leg0b["error"] and leg1b is None
