"""bench.py -- headline benchmark of the demod_2400 hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload sparse|dense]

A step = one pass of the hot path (icao_flush + to_mag + demodulate2400 per
131072-sample buffer, the unit of reference benches/demod_benchmark.rs:7-12) over one
256 MiB synthetic 2.4 MSPS i16 IQ buffer (512 buffers' worth) that is already resident
in HBM: scan kernel -> match kernel -> record kernel -> D2H of the trial records ->
ordered host replay -> ModeSMessage list on the host.  Nothing is skipped inside
the timed region.  The bench rotates over several distinct 256 MiB buffers so that a
step never re-reads data the 256 MiB Infinity Cache still holds.

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank demodulates its
own buffers as an independent stream -- the path shards by buffer with no data-path
collective (BASELINE.json north_star); torch.distributed is used for the barrier
and the max-over-ranks time only.  Weak scaling: per-GPU work is fixed.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

CHUNK = 131072
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak
BYTES_PER_SAMPLE = 4   # algorithmic bytes: one i16 IQ pair read per sample


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["sparse", "dense", "stream"], default="sparse",
                    help="sparse: 64 DF17 bursts per 256 MiB (BASELINE config 2); "
                         "dense: 5000 bursts (config 5); stream: host-resident IQ through the "
                         "pinned double-buffered ring, H2D inside the timed region (config 3)")
    ap.add_argument("--chunks", type=int, default=512, help="131072-sample buffers per step")
    ap.add_argument("--buffers", type=int, default=3, help="distinct IQ buffers rotated over")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timed-profiling", type=int, default=1,
                    help="HIP-event level inside the timed region (1 = scan kernel stamped by its "
                         "own launch; 0 = none, then roofline numbers come from the untimed repeat)")
    ap.add_argument("--depth", type=int, default=3,
                    help="passes in flight in the pipelined form (<= ADSB_MAX_IN_FLIGHT = 3): with 3 the "
                         "next scan is always queued on the device while the host collects")
    ap.add_argument("--sync", action="store_true",
                    help="one blocking adsb_demod_iq_device call per step instead of the two-deep "
                         "submit/collect pipeline")
    return ap.parse_args()


def main():
    args = parse()
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run "
                     "(one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: the demod_2400 path has no CPU fallback")
    # (ADSB_BENCH_BACKEND=gloo lets the N > 1 path be exercised on a box with fewer GPUs than
    # ranks: ranks then share devices and the timing reduction goes over CPU tensors)
    backend = os.environ.get("ADSB_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    from dump1090_rs_amd import Context, synth
    from dump1090_rs_amd._lib import AdsbMsg

    if args.workload == "stream":
        return stream_bench(args, torch, dist, rank, world, local_rank)

    n = args.chunks * CHUNK
    n_bursts = 64 if args.workload == "sparse" else 5000
    n_bursts = max(1, n_bursts * args.chunks // 512)
    dev = torch.device("cuda", local_rank)
    # distinct data per rank and per buffer: seed differs
    bufs = [synth.make_iq_torch(n, n_bursts=n_bursts, seed=synth.SEED_DEFAULT + 1000 * rank + b, device=dev)
            for b in range(args.buffers)]
    torch.cuda.synchronize()

    ctx = Context(device=local_rank, max_chunks=args.chunks)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream.cuda_stream)
    cap = 1 << 20
    out = (AdsbMsg * cap)()

    def run_steps(first: int, count: int, level: int):
        """`count` steps starting at step index `first`.  Returns (frames, summed stats).
        A step is icao_flush (benches/demod_benchmark.rs:9) + the whole pass over one buffer.
        Pipelined form: step i is submitted before step i-1 is collected, so the host part
        of one step (wait, copy-back, ordered replay) overlaps the device scan of the next;
        every step's full work still happens inside the loop."""
        ctx.set_profiling(level)
        tot = {"ms_scan": 0.0, "ms_scan_exclusive": 0.0, "ms_match": 0.0, "ms_records": 0.0, "ms_total_device": 0.0}
        frames = 0

        def account():
            st = ctx.stats_raw()
            for k in tot:
                tot[k] += getattr(st, k)

        for i in range(count):
            b = bufs[(first + i) % len(bufs)]
            ctx.icao_flush()
            if args.sync:
                frames += ctx.demod_iq_device_raw(b.data_ptr(), n, out, cap)
                account()
            else:
                ctx.submit_iq_device(b.data_ptr(), n)
                if i >= args.depth - 1:
                    frames += ctx.collect_raw(out, cap)
                    account()
        if not args.sync:
            for _ in range(min(count, args.depth - 1)):
                frames += ctx.collect_raw(out, cap)
                account()
        return frames, tot

    run_steps(0, args.warmup, 1)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Timed region: K steps with HIP events around the scan kernel only (level 1), recorded
    # by the library on the stream the kernels run on.
    fence()
    t0 = time.perf_counter()
    frames, tot = run_steps(args.warmup, args.steps, args.timed_profiling)
    fence()
    elapsed = time.perf_counter() - t0
    scan_ms = tot["ms_scan"]
    stats = ctx.stats()

    # untimed: the same steps once more with an event after every kernel, for the split
    _, tot2 = run_steps(args.warmup, args.steps, 2)
    if args.timed_profiling == 0:
        scan_ms = tot2["ms_scan"]
    match_ms, rec_ms, dev_ms = tot2["ms_match"], tot2["ms_records"], tot2["ms_total_device"]
    ctx.set_profiling(1)

    if dist is not None:
        from dump1090_rs_amd import sharding
        elapsed, frames = sharding.reduce_timing(dist, elapsed, frames, device=dev if backend == "nccl" else "cpu")

    total_samples = n * args.steps * world
    msps = total_samples / elapsed / 1e6
    scan_avg_s = scan_ms / args.steps / 1e3
    achieved = BYTES_PER_SAMPLE * n / scan_avg_s / 1e9 if scan_avg_s > 0 else 0.0

    result = {
        "metric": "IQ Msamples/s demodulated",
        "value": round(msps, 1),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "i16 IQ -> f32 magnitude (exact) -> u16/i32 integer",
        "data": "synthetic",
        "frames_per_s": round(frames / elapsed, 1),
        "frames_per_step": frames // max(1, args.steps * world),
        "config": {
            "workload": f"{args.chunks} x 131072-sample buffers = {n * 4 // (1 << 20)} MiB synthetic 2.4 MSPS "
                        f"i16 IQ resident in HBM, {n_bursts} injected Mode-S bursts ({args.workload}), "
                        f"icao_flush + to_mag + demodulate2400 per buffer, {args.buffers} distinct buffers rotated",
            "per_gpu_samples_per_step": n,
            "sharding": "independent stream per GPU, no collectives",
            "host_api": "blocking adsb_demod_iq_device per step" if args.sync else
                        f"adsb_submit_iq_device / adsb_collect, {args.depth} passes in flight",
            "kernels": "k_scan_fast (mag + sign planes + preamble + gates + trial syndromes) -> k_match -> k_records -> host replay",
            "library": "",
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": None,
            "kernel": "k_scan_fast",
            "kernel_avg_ms": round(scan_ms / args.steps, 4),
            "kernel_exclusive_avg_ms": round(tot["ms_scan_exclusive"] / args.steps, 4),
            "algorithmic_bytes_per_launch": BYTES_PER_SAMPLE * n,
            "other_kernels_avg_ms": {"k_match": round(match_ms / args.steps, 4),
                                     "k_records": round(rec_ms / args.steps, 4)},
            "device_chain_avg_ms": round(dev_ms / args.steps, 4),
            # consecutive pipelined launches overlap by about one tile round (two scan streams):
            # a launch's own duration then includes time it shared the GPU with its neighbour;
            # the rate the GPU sustains over whole steps is bytes / ms_per_step
            "launches_overlap": not args.sync,
            "achieved_over_steps": round(BYTES_PER_SAMPLE * n * args.steps / elapsed / 1e9, 1),
        },
        "device_stats_last_step": {k: stats[k] for k in
                                   ("n_candidates", "n_ap_entries", "n_records", "n_messages", "retries")},
    }
    from dump1090_rs_amd import _lib
    result["config"]["library"] = _lib.lib().adsb_version().decode()
    traffic_file = ROOT / "profiles" / "scan_hbm_traffic.json"
    if traffic_file.exists():
        try:
            tf = json.loads(traffic_file.read_text())
            if tf.get("library") == result["config"]["library"] and tf.get("chunks") == args.chunks:
                result["roofline"]["traffic"] = tf.get("bytes_per_launch")
                result["roofline"]["traffic_source"] = tf.get("source")
        except Exception:
            pass

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU baseline: the C restatement of the reference (oracle/, "port"), one thread,
        # on the host cores of this box, over buffer 0 of the same workload.  Also the
        # parity gate of this run: the GPU output for that buffer must be identical.
        from oracle import binding
        host = bufs[0].cpu().numpy()
        orc = binding.Oracle()
        orc.icao_flush()
        c0 = time.perf_counter()
        want, _ = orc.demod_iq(host, cap=cap)
        cpu_s = time.perf_counter() - c0
        # the same buffer over all host cores (workers per buffer + ordered replay)
        n_thr = max(1, min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), args.chunks))
        mt_s = None
        if n_thr > 1:
            orc_mt = binding.Oracle()
            orc_mt.icao_flush()
            orc_mt.demod_iq(host[: min(n, 16 * CHUNK)], cap=cap, threads=n_thr)  # spin the threads up once
            orc_mt.icao_flush()
            c0 = time.perf_counter()
            want_mt, _ = orc_mt.demod_iq(host, cap=cap, threads=n_thr)
            mt_s = time.perf_counter() - c0
            if want_mt != want:
                raise SystemExit("cpu_baseline: the multi-threaded oracle disagrees with the single-threaded one")
        ctx.icao_flush()
        got = ctx.demod_iq_device(bufs[0].data_ptr(), n, cap=cap)
        same = [(m.chunk, m.j, m.try_phase, m.score, m.msg, m.signal_level) for m in got] == \
               [(w["chunk"], w["j"], w["try_phase"], w["score"], w["msg"], w["signal_level"]) for w in want]
        result["cpu_baseline"] = {
            "value": round(n / cpu_s / 1e6, 2),
            "unit": "Msamples/s",
            "cores": 1,
            "kind": "port",
            "sample": f"buffer 0 of the workload, all {args.chunks} x 131072 samples once, {cpu_s:.2f} s; "
                      "C restatement of dump1090_rs (oracle/), not the Rust binary",
            "cpu": _cpu_model(),
            "host_cores_available": os.cpu_count(),
        }
        if mt_s:
            result["cpu_baseline"]["all_cores"] = {
                "value": round(n / mt_s / 1e6, 2), "unit": "Msamples/s", "cores": n_thr,
                "sample": f"the same buffer, {n_thr} threads over the 131072-sample buffers + ordered replay, {mt_s:.2f} s"}
        result["parity_checked"] = bool(same)
        result["parity_frames"] = len(want)
        if not same:
            result["parity_error"] = "GPU frame list differs from the CPU oracle"

    ctx.close()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and result.get("parity_checked") is False:
        sys.exit(3)


def stream_bench(args, torch, dist, rank, world, local_rank):
    """BASELINE config 3: sustained rate with the IQ starting in host memory.  A step = one
    ring slot of --chunks buffers (default here 64 = 32 MiB): adsb_ring_submit starts its
    pinned H2D copy on the copy stream and the pass behind it; the other slot's pass runs
    meanwhile.  No icao_flush between steps (the live loop of main.rs never flushes).  The
    ring buffers are filled once, outside the timed region (an SDR driver would DMA into them)."""
    from dump1090_rs_amd import Context, synth
    from dump1090_rs_amd._lib import AdsbMsg

    chunks = args.chunks if args.chunks != 512 else 64
    n = chunks * CHUNK
    ctx = Context(device=local_rank, max_chunks=chunks)
    ctx.ring_create(n)
    cap = 1 << 18
    out = (AdsbMsg * cap)()
    ctx.icao_flush()
    for k in range(2):  # fill both pinned slots (and warm up)
        buf = ctx.ring_acquire()
        buf[:] = synth.make_iq(n, n_bursts=max(1, 64 * chunks // 512), seed=synth.SEED_DEFAULT + 7 * rank + k)
        ctx.ring_submit(n)
        ctx.collect_raw(out, cap)
    for _ in range(args.warmup):
        ctx.ring_acquire()
        ctx.ring_submit(n)
        ctx.collect_raw(out, cap)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    ctx.set_profiling(1)
    frames, scan_ms = 0, 0.0
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ctx.ring_acquire()
        ctx.ring_submit(n)
        if i > 0:
            frames += ctx.collect_raw(out, cap)
            scan_ms += ctx.stats_raw().ms_scan
    frames += ctx.collect_raw(out, cap)
    scan_ms += ctx.stats_raw().ms_scan
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        from dump1090_rs_amd import sharding
        gloo = os.environ.get("ADSB_BENCH_BACKEND", "nccl") != "nccl"
        elapsed, frames = sharding.reduce_timing(dist, elapsed, frames, device="cpu" if gloo else torch.device("cuda", local_rank))
    from dump1090_rs_amd import _lib
    msps = n * args.steps * world / elapsed / 1e6
    scan_avg_s = scan_ms / args.steps / 1e3
    achieved = BYTES_PER_SAMPLE * n / scan_avg_s / 1e9 if scan_avg_s > 0 else 0.0
    result = {
        "metric": "IQ Msamples/s demodulated", "value": round(msps, 1), "unit": "Msamples/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "i16 IQ -> f32 magnitude (exact) -> u16/i32 integer", "data": "synthetic",
        "frames_per_s": round(frames / elapsed, 1),
        "config": {"workload": f"streaming ring: {chunks} x 131072-sample buffers = {n * 4 // (1 << 20)} MiB per slot, "
                               "host-resident IQ, pinned double-buffered hipMemcpyAsync inside the timed region "
                               "(BASELINE config 3)",
                   "h2d_GBps": round(BYTES_PER_SAMPLE * n * args.steps / elapsed / 1e9, 2),
                   "library": _lib.lib().adsb_version().decode()},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "kernel": "k_scan_fast",
                     "kernel_avg_ms": round(scan_ms / args.steps, 4),
                     "algorithmic_bytes_per_launch": BYTES_PER_SAMPLE * n,
                     "note": "PCIe-fed: the scan kernel idles between transfers; value is the sustained end-to-end rate"},
    }
    ctx.close()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
