"""K captures through adsb_multi_* on N contexts, for a kernel trace: tools/multi_overhead.sh runs this under
rocprofv3 --kernel-trace and tools/multi_overhead.py sets the kernels' union of intervals (what the device was
busy for) against the wall time of the same steps (printed here as JSON) -- the difference is what the
orchestration adds.  The timed steps sit between two 60 ms pauses so that the trace shows them as one cluster.
usage: python tools/multi_steps.py [--contexts 8] [--chunks 512] [--steps 40] [--pipelined]"""
import argparse, ctypes as C, json, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from dump1090_rs_amd import synth
from dump1090_rs_amd._lib import AdsbMsg
from dump1090_rs_amd.multi import MultiContext

ap = argparse.ArgumentParser()
ap.add_argument("--contexts", type=int, default=8)
ap.add_argument("--chunks", type=int, default=512, help="buffers in the whole capture")
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--pipelined", action="store_true")
ap.add_argument("--bursts", type=int, default=64, help="injected bursts per 512 buffers (64 = the sparse bench workload, 5000 = the dense one)")
ap.add_argument("--host", action="store_true", help="also time adsb_multi_demod_iq over the same capture in pageable host memory (the PCIe-inclusive rate)")
a = ap.parse_args()
CHUNK = 131072
n_dev = torch.cuda.device_count()
devices = [k % n_dev for k in range(a.contexts)]
multi = MultiContext(devices, -(-a.chunks // a.contexts))
ranges = multi.shard_ranges(a.chunks * CHUNK)
parts = [synth.make_iq_torch(n, n_bursts=max(1, a.bursts * (n // CHUNK) // 512), seed=synth.SEED_DEFAULT + 31 * r,
                             device=torch.device("cuda", devices[r])) for r, (_, n) in enumerate(ranges)]
torch.cuda.synchronize()
ptrs = (C.c_void_p * a.contexts)(*[C.c_void_p(t.data_ptr()) for t in parts])
ns = (C.c_size_t * a.contexts)(*[t.shape[0] for t in parts])
out = (AdsbMsg * (1 << 18))()
keys = ("ms_wall", "ms_phase1_max", "ms_phase2_max", "ms_phase1_span", "ms_phase2_span", "ms_exchange", "ms_replay")

def run(count, acc=None):
    done = 0
    for i in range(count):
        multi.icao_flush()
        multi.submit_raw(ptrs, ns)
        if not a.pipelined or i - done >= multi.max_in_flight() - 1:
            multi.collect_raw(out, 1 << 18)
            done += 1
            if acc is not None:
                st = multi.stats()
                for k in keys:
                    acc[k] += st[k]
    while done < count:
        multi.collect_raw(out, 1 << 18)
        done += 1

t = time.perf_counter()
while time.perf_counter() - t < 0.3:      # clocks up
    run(20)
time.sleep(0.06)
acc = {k: 0.0 for k in keys}
t0 = time.perf_counter()
run(a.steps, acc)
wall = time.perf_counter() - t0
time.sleep(0.06)
res = {"bursts_per_512_buffers": a.bursts, "last_capture": {k: multi.stats()[k] for k in ("n_records", "n_messages", "n_addrs_exchanged", "retries")}, "contexts": a.contexts, "devices": devices, "chunks": a.chunks, "steps": a.steps, "pipelined": a.pipelined,
       "ms_per_step_wall": round(wall / a.steps * 1e3, 4), "stats_mean": {k: round(v / a.steps, 4) for k, v in acc.items()}}
res["stats_mean"]["ms_overhead_host_clock"] = round(res["stats_mean"]["ms_wall"] - res["stats_mean"]["ms_phase1_span"] - res["stats_mean"]["ms_phase2_span"], 4)
if a.host:
    import numpy as np
    host = np.concatenate([t.cpu().numpy() for t in parts])
    multi.icao_flush()
    multi.demod_iq(host, cap=1 << 18)
    t0 = time.perf_counter()
    for _ in range(3):
        multi.icao_flush()
        multi.demod_iq(host, cap=1 << 18)
    dt = (time.perf_counter() - t0) / 3
    res["host_form"] = {"ms_per_capture": round(dt * 1e3, 3), "Msamples_per_s": round(len(host) / dt / 1e6, 1), "GB_per_s": round(4 * len(host) / dt / 1e9, 2),
                        "is": "adsb_multi_demod_iq from pageable host memory: every device thread copies its range to its device (hipMemcpyAsync through the runtime's staging) in front of its scan"}
    # ... and the asynchronous host form out of pinned memory (adsb_multi_host_alloc / adsb_multi_submit_iq): three
    # pinned captures rotated, four in flight -- every device's copy is a DMA over its own link, overlapped with the scans
    pinned = [multi.host_alloc(len(host)) for _ in range(3)]
    for p in pinned:
        p[:] = host
    def run_pinned(count):
        done = 0
        for i in range(count):
            multi.icao_flush()
            multi.submit_iq(pinned[i % 3])
            if i - done >= multi.max_in_flight() - 1:
                multi.collect_raw(out, 1 << 18)
                done += 1
        while done < count:
            multi.collect_raw(out, 1 << 18)
            done += 1
    run_pinned(4)
    t0 = time.perf_counter()
    run_pinned(12)
    dt = (time.perf_counter() - t0) / 12
    res["host_form_pinned"] = {"ms_per_capture": round(dt * 1e3, 3), "Msamples_per_s": round(len(host) / dt / 1e6, 1), "GB_per_s": round(4 * len(host) / dt / 1e9, 2),
                               "is": "adsb_multi_submit_iq out of adsb_multi_host_alloc memory, four captures in flight"}
    for p in pinned:
        multi.host_free(p)
print(json.dumps(res))
multi.close()
