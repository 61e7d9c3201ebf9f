"""The paths that only a box with several GPUs -- or the driver's `torch.distributed.run` -- ever takes, made to
run before that day.

* `bench.py` as ONE rank with the `nccl` backend (= RCCL) forced: process-group init with `device_id`, and every
  collective the bench issues (barrier in the fences, MAX / SUM all-reduce, all-gather of the per-rank step times,
  broadcast of rank 0's loop flag) on real hardware, on the one-GPU box of the driver's GPU suite.
* everything that takes a device index, on every device BUT 0: skipped (cleanly) on a one-GPU box, run wherever
  `torch.cuda.device_count() >= 2` -- a context, the ring, a registered host buffer, the shard phases from a
  worker thread, two devices at once from two threads -- each against the CPU oracle.
"""
import json
import os
import socket
import subprocess
import sys
import threading

import numpy as np
import pytest

from dump1090_rs_amd import synth
from tests.conftest import ROOT

pytestmark = pytest.mark.gpu

CHUNK = 131072


def key(m):
    return (m.chunk, m.j, m.try_phase, m.score, m.msg, m.signal_level)


def want_key(w):
    return (w["chunk"], w["j"], w["try_phase"], w["score"], w["msg"], w["signal_level"])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench_as_one_nccl_rank(*extra, timeout=600, also=False):
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", LOCAL_WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), ADSB_BENCH_FORCE_DIST="1", ADSB_BENCH_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--blocks", "1",
                        "--ramp-ms", "20", *([] if also else ["--no-also"]), *extra], capture_output=True, text=True, timeout=timeout, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_as_one_rank_over_rccl(hip_lib, oracle_mod):
    """`init_process_group("nccl", device_id=...)` and the bench's fence / reduce / max_each / gather / any_rank0
    over RCCL at world size 1: the code the driver's multi-GPU run executes, today.  The line still verifies itself
    (rank 0's buffer 0 against the oracle) and says which backend carried it."""
    line = _bench_as_one_nccl_rank("--chunks", "64")
    assert line["backend"] == "nccl" and line["world_size_seen"] == 1 and line["n_gpus"] == 1
    assert line["value"] > 0 and line["steps"] == 4 and line["parity_checked"] is True
    assert len(line["per_rank_ms_per_step"]) == 1 and line["per_rank_ms_per_step"][0] > 0
    assert line["ms_per_step_blocks"] and len(line["ms_per_step_blocks"]["all"]) == 1     # max_each went through all_reduce
    assert line["config"]["clock_ramp"]["steps"] >= 20                                     # any_rank0 broadcast ended the ramp
    assert "roofline" in line and "cpu_baseline" in line


def test_bench_shard_workload_as_one_rank_over_rccl(hip_lib, oracle_mod):
    """... and `--workload shard` (BASELINE config 4: one capture cut into ranges, the merge checked against the
    single stream) with RCCL as the default backend."""
    line = _bench_as_one_nccl_rank("--workload", "shard", "--capture-chunks", "48")
    assert line["backend"] == "nccl" and line["shard_merge_equals_single_stream"] is True and line["scaling"] == "strong"
    assert line["parity_checked"] is True and line["roofline"]["bound"] == "hbm" and 0 < line["roofline"]["frac"] < 1     # (against the oracle too)


def test_bench_one_process_over_all_devices_behind_the_ranks_timed_region(hip_lib, oracle_mod):
    """What `bench.py --gpus N` does after its timed independent-stream region: every rank closes
    its context and waits at a host-side (gloo) barrier while rank 0 drives ONE adsb_multi over all the devices it sees
    -- the reference's shape, one process and one filter (dump1090_rs/src/main.rs:154-167) -- sparse and busy sky, each
    checked against the threaded oracle over the whole capture.  Here: one rank, RCCL as the default backend, a
    capture of 96 buffers."""
    line = _bench_as_one_nccl_rank("--chunks", "32", "--capture-chunks", "96", also=True)
    assert line["backend"] == "nccl" and line["parity_checked"] is True
    leg = line["also"]["config4_one_process_n_devices"]
    assert leg["parity_checked"] is True and "error" not in leg
    assert [r["sky"] for r in leg["runs"]] == ["sparse", "busy_sky"]
    for r in leg["runs"]:
        assert r["parity_checked"] is True and r["parity_frames"] > 0 and r["value"] > 0 and r["devices"] == [0]
        assert r["roofline"]["bound"] == "hbm" and 0 < r["roofline"]["frac"] < 1
        assert set(r["blocking_steps_host_clock"]) >= {"ms_wall", "ms_phase1_span", "ms_phase2_span", "ms_exchange", "ms_replay", "ms_overhead"}
    assert set(line["also"]) == {"config4_one_process_n_devices"}     # (the N = 1 legs are not run under a process group)


# ------------------------------------------------------------------------------------------------ device > 0
def other_devices():
    import torch
    return list(range(1, torch.cuda.device_count()))


def needs_two_devices():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one HIP device on this box: the device-index paths beyond 0 run where there are two or more")


def test_a_context_on_every_other_device_equals_the_oracle(hip_lib, oracle_mod):
    """adsb_create(device = k > 0): blocking host / device-resident calls, the pipelined form, a large pass and
    passes of one buffer -- every launch must land on the context's device whatever the calling thread's is."""
    needs_two_devices()
    import torch
    from dump1090_rs_amd import Context
    iq = synth.make_iq(20 * CHUNK + 4321, n_bursts=200, seed=8801, n_icao=10, df11_every=4)
    want, _ = oracle_mod.Oracle().demod_iq(iq)
    one = iq[3 * CHUNK:4 * CHUNK]
    want_one, _ = oracle_mod.Oracle().demod_iq(one)
    for dev in other_devices():
        torch.cuda.set_device(0)                      # (the caller's current device is NOT the context's)
        resident = torch.from_numpy(iq).to(f"cuda:{dev}")
        torch.cuda.synchronize(dev)
        with Context(dev, 32) as c:
            c.icao_flush()
            assert [key(m) for m in c.demod_iq(iq)] == [want_key(w) for w in want]
            c.icao_flush()
            assert [key(m) for m in c.demod_iq_device(resident.data_ptr(), len(iq))] == [want_key(w) for w in want]
            for _ in range(3):
                c.icao_flush()
                c.submit_iq_device(resident.data_ptr(), len(iq))
            for _ in range(3):
                assert [key(m) for m in c.collect()] == [want_key(w) for w in want]
        with Context(dev, 1) as c:
            for _ in range(20):
                c.icao_flush()
                assert [key(m) for m in c.demod_iq(one)] == [want_key(w) for w in want_one]
            mag = c.to_mag(one)
            c.icao_flush()
            assert [key(m) for m in c.demodulate2400(mag)] == [want_key(w) for w in want_one]


def test_the_ring_and_a_registered_host_buffer_on_every_other_device(hip_lib, oracle_mod):
    """Pinned, mapped host memory is mapped for the context's device: the ring's slots read in place by device k,
    the caller's own registered buffer, slots copied in front of their pass."""
    needs_two_devices()
    from dump1090_rs_amd import Context
    iq = synth.make_iq(24 * CHUNK, n_bursts=240, seed=8802, n_icao=12, df11_every=5)
    want, _ = oracle_mod.Oracle().demod_iq(iq)
    for dev in other_devices():
        for per_slot in (1, 4):
            with Context(dev, per_slot) as c:
                c.ring_create(per_slot * CHUNK)
                c.icao_flush()
                got = []
                n_slots = 24 // per_slot
                for b in range(n_slots):
                    if c.pending() == c.max_in_flight():
                        got += [m for m in c.collect()]
                    buf = c.ring_acquire()
                    buf[:] = iq[b * per_slot * CHUNK:(b + 1) * per_slot * CHUNK]
                    c.ring_submit(per_slot * CHUNK)
                while c.pending():
                    got += [m for m in c.collect()]
                # (chunk is the buffer's index within its slot: compare everything else, in order)
                assert [key(m)[1:] for m in got] == [want_key(w)[1:] for w in want]
        own = np.ascontiguousarray(iq[: 2 * CHUNK]).copy()
        want_own, _ = oracle_mod.Oracle().demod_iq(own)
        with Context(dev, 2) as c:
            c.host_register(own)
            c.icao_flush()
            assert [key(m) for m in c.demod_iq(own)] == [want_key(w) for w in want_own]
            c.host_unregister(own)


def test_shard_phases_from_a_worker_thread_on_every_other_device(hip_lib, oracle_mod):
    """adsb_shard_scan / adsb_shard_finish called from threads that never touched HIP (their current device is 0)
    on a context of device k: the entry points select the context's device themselves."""
    needs_two_devices()
    import torch
    from dump1090_rs_amd import Context
    from dump1090_rs_amd.context import replay_records
    iq = synth.make_iq(5 * CHUNK - 777, n_bursts=60, seed=8803, n_icao=6, df11_every=4)
    want, _ = oracle_mod.Oracle().demod_iq(iq)
    for dev in other_devices():
        resident = torch.from_numpy(iq).to(f"cuda:{dev}")
        torch.cuda.synchronize(dev)
        box = {}
        with Context(dev, 8) as c:
            c.icao_flush()
            for fn in (lambda: box.__setitem__("learned", c.shard_scan(resident.data_ptr(), len(iq))),
                       lambda: box.__setitem__("records", c.shard_finish(box["learned"]))):
                t = threading.Thread(target=fn)
                t.start()
                t.join(120)
                assert not t.is_alive()
            assert [key(m) for m in replay_records(box["records"])] == [want_key(w) for w in want]


def test_every_device_at_once_from_its_own_thread(hip_lib, oracle_mod):
    """One host thread and one context per device, all demodulating at the same time (each its own stream of
    buffers, its own filter): what `bench.py --gpus N` does with processes, with threads."""
    needs_two_devices()
    import torch
    from dump1090_rs_amd import Context
    n_dev = torch.cuda.device_count()
    iqs = [synth.make_iq(12 * CHUNK, n_bursts=100, seed=8810 + d, n_icao=8) for d in range(n_dev)]
    wants = [[want_key(w) for w in oracle_mod.Oracle().demod_iq(iq)[0]] for iq in iqs]
    errors = []

    def worker(d):
        try:
            with Context(d, 12) as c:
                for _ in range(5):
                    c.icao_flush()
                    got = [key(m) for m in c.demod_iq(iqs[d])]
                    if got != wants[d]:
                        errors.append((d, "differs"))
        except Exception as e:   # noqa: BLE001
            errors.append((d, repr(e)))

    threads = [threading.Thread(target=worker, args=(d,)) for d in range(n_dev)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(600)
    assert not errors and not any(t.is_alive() for t in threads), errors
