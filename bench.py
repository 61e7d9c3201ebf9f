"""bench.py -- headline benchmark of the demod_2400 hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload sparse|dense|stream|shard]

A step = one pass of the hot path over one 256 MiB synthetic 2.4 MSPS i16 IQ buffer (512
131072-sample buffers) that is already resident in HBM: ONE icao_flush per step, then what the
reference does per buffer (to_mag + demodulate2400, benches/demod_benchmark.rs:10-11) for all 512
-- scan kernel -> match -> order -> records -> ordered host replay -> ModeSMessage list on the host.
Nothing is skipped inside the timed region.  The bench rotates over several distinct 256 MiB
buffers so that a step never re-reads data the 256 MiB Infinity Cache still holds.

--gpus N > 1: one rank per GPU.  A plain `python bench.py --gpus N` starts the N ranks itself
(torch.distributed.run, before this process has touched a GPU); under torch.distributed.run it
is simply rank RANK.  Every rank demodulates its own buffers as an independent stream -- the path
shards by buffer with no data-path collective (BASELINE.json north_star); torch.distributed is
used for the barrier and the max-over-ranks time only.  Weak scaling: per-GPU work is fixed.
--workload shard (BASELINE config 4): ONE capture of N x 512 buffers cut into contiguous ranges,
one per rank; the merged frame list is checked against the single-stream result.

At N = 1 the line also carries, under "also", short legs for the other BASELINE configs
(1: the reference's `cargo bench` case, 3: streaming ring, 5: dense input), each with its own
parity flag.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

# (No GPU_MAX_HW_QUEUES here any more: the library keeps every context's streams within four per priority and device --
# dump1090_rs_amd/csrc/adsb_context.cpp: DeviceStreams -- so a process that mixes a large context with contexts for
# passes of a few buffers, as this one does in its `also` legs, needs nothing in its environment.  A value the caller
# has set is left alone and reported under config.runtime_env.)

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

CHUNK = 131072
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak
HBM_ACHIEVABLE_GBS = 6290.0  # ... and what a float4 copy measures on the chip (79 % of it): the guide's "achievable"
BYTES_PER_SAMPLE = 4   # algorithmic bytes: one i16 IQ pair read per sample
PUBLISHED_CONFIG1_MS = 3.6950  # reference README.md:107, bench "01", Intel i7-7700K, 1 thread
DTYPE = "i16 IQ -> f32 magnitude (exact) -> u16/i32 integer"


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["sparse", "dense", "stream", "shard", "live"], default="sparse",
                    help="sparse: 64 DF17 bursts per 256 MiB (BASELINE config 2); "
                         "dense: 5000 bursts (config 5); stream: host-resident IQ through the "
                         "pinned double-buffered ring, H2D inside the timed region (config 3); "
                         "shard: one capture of N x --chunks buffers cut into contiguous ranges (config 4); "
                         "live: the receiver's loop, one 131072-sample buffer per pass through the ring, the filter "
                         "never flushed (main.rs:154-167), checked against one oracle stream over 160 distinct passes")
    ap.add_argument("--chunks", type=int, default=512, help="131072-sample buffers per step (and per GPU)")
    ap.add_argument("--buffers", type=int, default=3, help="distinct IQ buffers rotated over")
    ap.add_argument("--capture-chunks", type=int, default=4096,
                    help="--workload shard: buffers in the ONE capture that is cut into N contiguous ranges "
                         "(BASELINE config 4: 8 x the 256 MiB buffer = 2 GiB = 4096; the total is fixed, so the "
                         "line says scaling: strong)")
    ap.add_argument("--ramp-ms", type=float, default=120.0,
                    help="untimed passes of the same workload before the W warm-up steps, for this long: the "
                         "GPU raises its clocks over ~50 ms of sustained load (a pass is 0.1 ms), and a stream "
                         "demodulator's throughput is what it sustains, not what the first 25 passes after "
                         "idling make; 0 = none.  Reported under config.clock_ramp with the cold step time.")
    ap.add_argument("--blocks", type=int, default=5,
                    help="further blocks of --steps steps after the timed one, each timed the same way: the line's "
                         "ms_per_step_blocks (min / median / max); the headline stays the first block")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pin", action="store_true",
                    help="leave the rank's CPU affinity alone (default: the cores of its GPU's NUMA node, from sysfs)")
    ap.add_argument("--stream-seconds", type=float, default=10.0,
                    help="--workload stream and the config-3 leg: run at least this long at the headline slot size")
    ap.add_argument("--no-also", action="store_true", help="skip the short legs for BASELINE configs 1, 3, 4 and 5")
    ap.add_argument("--also-only", default="",
                    help="comma-separated subset of the `also` legs to run (config1, config3, live, config4, config5); default: all")
    ap.add_argument("--timed-profiling", type=int, default=1,
                    help="HIP-event level inside the timed region (1 = scan kernel stamped by its "
                         "own launch; 0 = none, then roofline numbers come from the untimed repeat)")
    ap.add_argument("--depth", type=int, default=4,
                    help="passes in flight in the pipelined form (<= ADSB_MAX_IN_FLIGHT = 4): from 3 on the next "
                         "scan is always queued on the device while the host collects; a dense stream, whose "
                         "passes are ordered and scored by six more small kernels, needs the fourth")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher self-test (no GPU, no measurement): the ranks rendezvous over gloo, "
                         "barrier, and rank 0 prints a line that says so")
    ap.add_argument("--single-process", action="store_true",
                    help="--workload shard: ONE process drives all devices through adsb_multi_* (a context and a host "
                         "thread per device inside the library, the address exchange in memory, one replay) instead "
                         "of one rank per GPU; --contexts picks how many shards")
    ap.add_argument("--contexts", type=int, default=0,
                    help="--single-process: shards of the capture (default: every visible device once); more shards than "
                         "devices wrap around (8 on a one-GPU box = eight contexts sharing the GPU)")
    ap.add_argument("--sync", action="store_true",
                    help="one blocking adsb_demod_iq_device call per step instead of the "
                         "submit/collect pipeline")
    return ap.parse_args(argv)


def self_launch(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N ranks under torch.distributed.run.
    Nothing in THIS process has touched a GPU (torch is not even imported), it only waits for the
    children and hands rank 0's JSON line on."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if line is not None:
        print(line, flush=True)
    elif proc.stdout:
        sys.stderr.write(proc.stdout)
    return proc.returncode if proc.returncode else (0 if line is not None else 1)


def _median(xs):
    s = sorted(xs)
    return s[len(s) // 2] if len(s) % 2 else 0.5 * (s[len(s) // 2 - 1] + s[len(s) // 2])


def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_quota():
    """CPUs' worth of time the container's cgroup allows (cpu.max / cfs quota), or None when unlimited:
    os.cpu_count() counts what the machine has, not what this process may use."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def _same(got, want) -> bool:
    return [(m.chunk, m.j, m.try_phase, m.score, m.msg, m.signal_level) for m in got] == \
           [(w["chunk"], w["j"], w["try_phase"], w["score"], w["msg"], w["signal_level"]) for w in want]


class Env:
    """Rank / device / process group of this process."""

    def __init__(self, args):
        import torch
        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # Before anything touches the GPU: this rank onto the cores of its GPU's NUMA node (a step is
        # ~60 us of HIP calls on one host thread out of ~95; eight ranks left to the scheduler on a
        # two-socket box contend and cross the socket link).  Reads sysfs only.
        from dump1090_rs_amd import sharding
        self.cpus_before = sorted(os.sched_getaffinity(0))
        self.affinity = None
        if not args.no_pin:
            self.affinity = sharding.pin_to_gpu_numa_node(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(self.world))))
        if not torch.cuda.is_available():
            sys.exit("bench.py needs a GPU: the demod_2400 path has no CPU fallback")
        # (ADSB_BENCH_BACKEND=gloo lets the N > 1 path be exercised on a box with fewer GPUs than
        # ranks: ranks then share devices and the timing reduction goes over CPU tensors)
        self.backend = os.environ.get("ADSB_BENCH_BACKEND", "nccl")
        if self.backend != "nccl":
            local_rank = local_rank % max(1, torch.cuda.device_count())
        self.local_rank = local_rank
        torch.cuda.set_device(local_rank)
        self.dev = torch.device("cuda", local_rank)
        self.dist = None
        # (ADSB_BENCH_FORCE_DIST=1: a single rank still joins a process group and every fence / reduction below goes
        # through the backend's collectives -- how the RCCL path of this file is exercised on a one-GPU box,
        # tests/test_gpu_multidevice.py)
        if self.world > 1 or os.environ.get("ADSB_BENCH_FORCE_DIST"):
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=self.dev)
            else:
                dist.init_process_group(backend=self.backend)
            self.dist = dist
        self.reduce_device = self.dev if self.backend == "nccl" else "cpu"

    def fence(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def any_rank0(self, flag: bool) -> bool:
        """rank 0's flag on every rank (loops that contain a barrier must end together)"""
        if self.dist is None:
            return flag
        t = self.torch.tensor([1 if flag else 0], dtype=self.torch.int32, device=self.reduce_device)
        self.dist.broadcast(t, src=0)
        return bool(t.item())

    def reduce(self, elapsed, frames):
        if self.dist is None:
            return elapsed, frames
        from dump1090_rs_amd import sharding
        return sharding.reduce_timing(self.dist, elapsed, frames, device=self.reduce_device)

    def max_each(self, values):
        """element-wise MAX over ranks of a list of floats (a block takes as long as its slowest rank)"""
        if self.dist is None or not values:
            return list(values)
        t = self.torch.tensor(values, dtype=self.torch.float64, device=self.reduce_device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(x) for x in t.cpu().tolist()]

    def gather(self, value: float):
        """One float per rank, on every rank (per-rank step times: a straggler must not hide behind the MAX)."""
        if self.dist is None:
            return [value]
        t = self.torch.tensor([value], dtype=self.torch.float64, device=self.reduce_device)
        out = self.torch.zeros(self.world, dtype=self.torch.float64, device=self.reduce_device)
        self.dist.all_gather_into_tensor(out, t)
        return [float(x) for x in out.cpu().tolist()]

    def all_cores(self):
        """Context manager: the affinity this process started with (the CPU baseline uses every core)."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            now = os.sched_getaffinity(0)
            try:
                os.sched_setaffinity(0, self.cpus_before)
                yield
            finally:
                os.sched_setaffinity(0, now)
        return cm()

    def affinity_summary(self):
        a = self.affinity
        if not a:
            return None
        return {"numa_node": a["numa_node"], "cpus": len(a["cpus"]), "first_cpu": a["cpus"][0] if a["cpus"] else None,
                "cpus_before": a["cpus_before"], "ranks_on_node": a["ranks_on_node"], "gpu": a["gpu"],
                "applied": a["applied"], "source": a["source"]}

    def finish(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def run_resident(env: Env, args, workload: str, steps: int, warmup: int, level2: bool = True, short_block: int = 0):
    """The device-resident pass loop (BASELINE configs 2 and 5).  Returns a dict of raw numbers and
    keeps the context / buffers alive in it for the parity leg."""
    torch = env.torch
    from dump1090_rs_amd import Context, synth
    from dump1090_rs_amd._lib import AdsbMsg

    n = args.chunks * CHUNK
    n_bursts = 64 if workload == "sparse" else 5000
    n_bursts = max(1, n_bursts * args.chunks // 512)
    # distinct data per rank and per buffer: the seed differs
    bufs = [synth.make_iq_torch(n, n_bursts=n_bursts, seed=synth.SEED_DEFAULT + 1000 * env.rank + b, device=env.dev)
            for b in range(args.buffers)]
    torch.cuda.synchronize()
    ctx = Context(device=env.local_rank, max_chunks=args.chunks)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    cap = 1 << 20
    out = (AdsbMsg * cap)()
    depth = max(1, min(4, args.depth))

    def run_steps(first: int, count: int, level: int, blocking: bool = args.sync):
        """`count` steps starting at step index `first`.  A step is ONE icao_flush
        (benches/demod_benchmark.rs:9 flushes per call) + the whole pass over one 256 MiB buffer.
        Pipelined form: step i is submitted before step i - (depth-1) is collected, so the host part
        of one step (wait, ordered replay) overlaps the device scans of the next; every step's full
        work still happens inside the loop.  Returns frames, summed stats, collect timestamps."""
        ctx.set_profiling(level)
        tot = {"ms_scan": 0.0, "ms_scan_exclusive": 0.0, "ms_match": 0.0, "ms_records": 0.0, "ms_total_device": 0.0}
        frames, stamps = 0, []

        def account():
            st = ctx.stats_raw()
            for k in tot:
                tot[k] += getattr(st, k)
            stamps.append(time.perf_counter())

        for i in range(count):
            b = bufs[(first + i) % len(bufs)]
            ctx.icao_flush()
            if blocking:
                frames += ctx.demod_iq_device_raw(b.data_ptr(), n, out, cap)
                account()
            else:
                ctx.submit_iq_device(b.data_ptr(), n)
                if i >= depth - 1:
                    frames += ctx.collect_raw(out, cap)
                    account()
        if not blocking:
            for _ in range(min(count, depth - 1)):
                frames += ctx.collect_raw(out, cap)
                account()
        return frames, tot, stamps

    # (a dense stream switches to device-side ordering and scoring once the context has seen how dense it
    # is, and finishes the passes then in flight early to rebuild the device's copy of the filter: that
    # transition belongs to the warm-up, not to the steady state being timed)
    run_steps(0, 6 if workload == "dense" else 0, 1)
    # (no collector pause inside a 2 ms timed region -- and none between the ramp and it either: the
    # GPU must not idle there, a full collection takes milliseconds)
    gc.collect()
    gc.disable()
    ramp = clock_ramp(env, args, lambda first, count: run_steps(first, count, 1))
    run_steps(0, warmup, 1)
    # Timed region: K steps with HIP events around the scan kernel only (level 1), stamped by the
    # scan launch itself on the stream it runs on.
    env.fence()
    t0 = time.perf_counter()
    frames, tot, stamps = run_steps(warmup, steps, args.timed_profiling)
    env.fence()
    elapsed = time.perf_counter() - t0
    # The spread: the same K steps again, block after block, each between the same fences.  Not the
    # headline (that is the block above, as the driver's contract wants it) -- what one 2 ms sample cannot
    # say about itself.
    blocks = []
    for b in range(max(0, args.blocks)):
        env.fence()
        tb = time.perf_counter()
        run_steps(warmup + steps * (b + 1), steps, args.timed_profiling)
        env.fence()
        blocks.append((time.perf_counter() - tb) / steps * 1e3)
    short_ms = None
    if short_block > 0:   # one block of fewer steps between the same fences: what the pipeline's fill and drain weigh in it
        env.fence()
        tb = time.perf_counter()
        run_steps(warmup, short_block, args.timed_profiling)
        env.fence()
        short_ms = (time.perf_counter() - tb) / short_block * 1e3
    gc.enable()
    stats = ctx.stats()
    tot2 = None
    if level2:  # untimed: the same steps once more with an event after every kernel, for the split
        _, tot2, _ = run_steps(warmup, steps, 2)
        if args.timed_profiling == 0:
            tot = tot2
    # untimed: the same steps with blocking calls, where launches do not overlap and a launch's own
    # start-to-stop time is the kernel alone (what `bench.py --sync` and profiles/*_sync_* report)
    alone_ms = None
    if level2 and not args.sync:
        run_steps(warmup, 30, 1, blocking=True)   # (the switch from pipelined to blocking calls settles first)
        _, tot3, _ = run_steps(warmup, steps, 1, blocking=True)
        alone_ms = tot3["ms_scan"] / steps
    ctx.set_profiling(1)
    # step-to-step intervals between consecutive collects (steady state of the pipeline)
    iv = [b - a for a, b in zip(stamps, stamps[1:])]
    return {"ctx": ctx, "bufs": bufs, "n": n, "n_bursts": n_bursts, "cap": cap, "frames": frames, "elapsed": elapsed,
            "tot": tot, "tot2": tot2, "stats": stats, "intervals": iv, "depth": depth, "ramp": ramp,
            "alone_ms": alone_ms, "blocks": blocks, "short_block_ms": short_ms}


def clock_ramp(env: Env, args, run) -> dict:
    """Untimed passes of the workload until --ramp-ms have gone by (in batches of 20 steps): after
    idling the chip starts a load at reduced clocks and takes tens of milliseconds of sustained work
    to reach the ones it then holds (measured: step and scan-kernel time fall by ~20 % over the first
    ~400 passes, then stay).  The timed region that follows is still exactly K steps after W warm-up
    steps; this is what makes it the steady state.  Returns what was done, with the cold step time."""
    out = {"ms": args.ramp_ms, "steps": 0, "cold_ms_per_step": None, "last_ms_per_step": None}
    if args.ramp_ms <= 0:
        return out
    env.fence()
    t0 = time.perf_counter()
    while True:
        t1 = time.perf_counter()
        run(out["steps"], 20)
        env.fence()
        dt = (time.perf_counter() - t1) / 20 * 1e3
        if out["steps"] == 0:
            out["cold_ms_per_step"] = round(dt, 4)
        out["last_ms_per_step"] = round(dt, 4)
        out["steps"] += 20
        # every rank leaves the loop after the same batch (the ranks' clocks differ: rank 0 decides)
        if not env.any_rank0((time.perf_counter() - t0) * 1e3 < args.ramp_ms) or out["steps"] >= 20000:
            break
    return out


def parity_leg(env: Env, r, chunks: int, baseline: bool = True):
    """The parity gate -- the GPU frame list of buffer 0 must be identical to the CPU oracle's -- and,
    with `baseline`, the CPU baseline on that buffer (the C restatement of the reference, 1 thread and
    all threads, built -O3 -march=native on this box for the timing)."""
    from oracle import binding
    ctx, bufs, n, cap = r["ctx"], r["bufs"], r["n"], r["cap"]
    host = bufs[0].cpu().numpy()
    orc = binding.Oracle()
    orc.icao_flush()
    c0 = time.perf_counter()
    want, _ = orc.demod_iq(host, cap=cap)
    cpu_s = time.perf_counter() - c0
    ctx.icao_flush()
    got = ctx.demod_iq_device(bufs[0].data_ptr(), n, cap=cap)
    same = _same(got, want)
    if not baseline:
        return None, same, len(want)
    built = "-O3 -march=x86-64-v3 (the checker's own build)"
    native = binding.build_native()
    L = None
    if native is not None:
        L = binding.load(native)
        t = binding.Oracle(L)
        t.icao_flush()
        c0 = time.perf_counter()
        want_n, _ = t.demod_iq(host, cap=cap)
        native_s = time.perf_counter() - c0
        if want_n != want:
            raise SystemExit("cpu_baseline: the -march=native build of the oracle disagrees with the checker's build")
        cpu_s, built = native_s, "-O3 -march=native on this box"
    mt_s, n_thr = None, 1
    with env.all_cores():   # (the rank itself is pinned to its GPU's NUMA node; the baseline gets the whole box)
        n_thr = max(1, min(os.cpu_count() or 1, len(os.sched_getaffinity(0)), chunks))
        # (a container with a CPU quota: threads beyond twice the quota only get throttled -- measured on a box
        # that shows 256 CPUs and grants ~16: 32 threads 59 ms per 256 MiB, 256 threads 93 ms)
        quota = _cpu_quota()
        if quota:
            n_thr = max(1, min(n_thr, int(2 * quota + 0.5)))
        if n_thr > 1:
            orc_mt = binding.Oracle(L)
            orc_mt.icao_flush()
            orc_mt.demod_iq(host[: min(n, 16 * CHUNK)], cap=cap, threads=n_thr)  # spin the threads up once
            times = []
            for _ in range(3):   # (tens of milliseconds each: median of three)
                orc_mt.icao_flush()
                want_mt, _ = orc_mt.demod_iq(host, cap=cap, threads=n_thr, timing=times)
                if want_mt != want:
                    raise SystemExit("cpu_baseline: the multi-threaded oracle disagrees with the single-threaded one")
            mt_s = _median(times)
    base = {
        "value": round(n / cpu_s / 1e6, 2), "unit": "Msamples/s", "cores": 1, "kind": "port",
        "sample": f"buffer 0 of the workload, all {chunks} x 131072 samples once, {cpu_s:.2f} s; "
                  f"C restatement of dump1090_rs (oracle/), not the Rust binary; built {built}",
        "cpu": _cpu_model(), "host_cores_available": os.cpu_count(),
    }
    if mt_s:
        base["all_cores"] = {"value": round(n / mt_s / 1e6, 2), "unit": "Msamples/s", "cores": n_thr,
                             "cpu_quota_cores": quota,
                             "sample": f"the same buffer, {n_thr} threads: workers run to_mag + gates + slicer + DF / CRC class "
                                       f"per 131072-sample buffer, only trials that can score or add reach the serial ordered "
                                       f"replay (oracle/dump1090_oracle_mt.c: its serial stage is microseconds); {mt_s * 1e3:.1f} ms "
                                       "inside the C call, median of three; equal to the one-thread oracle.  cores = threads "
                                       "used; cpu_quota_cores = what the container's cgroup grants (null: no limit) -- the box "
                                       "shows 256 CPUs"}
    if native is not None:
        try:
            os.unlink(native)
        except OSError:
            pass
    return base, same, len(want)


def _profile_json(name: str, library: str, chunks: int):
    """A committed measurement under profiles/ for THIS library build and workload size, or None."""
    f = ROOT / "profiles" / name
    try:
        d = json.loads(f.read_text())
    except (OSError, ValueError):
        return None
    return d if d.get("library") == library and d.get("chunks", chunks) == chunks else None


def resident_result(env: Env, args, r, workload: str):
    from dump1090_rs_amd import _lib
    n, steps = r["n"], args.steps
    per_rank_ms = env.gather(r["elapsed"] / steps * 1e3)
    elapsed, frames = env.reduce(r["elapsed"], r["frames"])
    blocks = env.max_each(r.get("blocks") or [])
    tot, tot2 = r["tot"], r["tot2"] or r["tot"]
    own_ms = tot["ms_scan"] / steps
    # Consecutive pipelined scans overlap (the next one's workgroups fill the CUs as the previous grid
    # drains, and with several passes in flight the next launch is dispatched while its predecessor still
    # runs): a launch's own start-to-stop time then counts the shared stretch twice.  The device time
    # per launch is the union of the launches' intervals / launches = ms_scan_exclusive.
    excl_ms = tot["ms_scan_exclusive"] / steps
    # roofline.achieved = algorithmic bytes per launch / the kernel's average launch duration, where a
    # launch's duration is its own: launches that do NOT overlap (blocking calls), measured here with the
    # HIP events the launch stamps on its stream -- the figure `rocprofv3 --kernel-trace --stats` gives
    # for `bench.py --sync` (profiles/*_sync_kernel_stats.csv).  The pipelined figures (what a stream
    # sustains) are under roofline.sustained.
    alone_ms = own_ms if args.sync else r.get("alone_ms")
    kernel_ms = alone_ms if alone_ms else excl_ms
    algo = BYTES_PER_SAMPLE * n
    gbs = lambda ms: algo / (ms / 1e3) / 1e9 if ms and ms > 0 else 0.0
    achieved = gbs(kernel_ms)
    iv = r["intervals"]
    library = _lib.lib().adsb_version().decode()
    result = {
        "metric": "IQ Msamples/s demodulated",
        "value": round(n * steps * env.world / elapsed / 1e6, 1),
        "unit": "Msamples/s",
        "n_gpus": env.world,
        "steps": steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 4),
        "ms_per_step_median": round(_median(iv) * 1e3, 4) if iv else None,
        "ms_per_step_cold": r["ramp"]["cold_ms_per_step"],
        "ms_per_step_blocks": None if not blocks else {
            "is": f"{len(blocks)} further blocks of {steps} steps right after the timed one, each timed like it "
                  "(fence, K steps, fence; MAX over ranks): the spread a single block cannot show",
            "min": round(min(blocks), 4), "median": round(_median(blocks), 4), "max": round(max(blocks), 4),
            "all": [round(x, 4) for x in blocks]},
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": DTYPE,
        "data": "synthetic",
        "frames_per_s": round(frames / elapsed, 1),
        "frames_per_step": frames // max(1, steps * env.world),
        "per_rank_ms_per_step": [round(x, 4) for x in per_rank_ms],
        "backend": env.backend if env.dist is not None else None,
        "world_size_seen": env.world,
        "config": {
            "workload": f"{args.chunks} x 131072-sample buffers = {n * 4 // (1 << 20)} MiB synthetic 2.4 MSPS "
                        f"i16 IQ resident in HBM, {r['n_bursts']} injected Mode-S bursts ({workload}); a step = one "
                        f"icao_flush + (to_mag + demodulate2400 per buffer) over all {args.chunks} buffers, "
                        f"{args.buffers} distinct 256 MiB buffers rotated",
            "per_gpu_samples_per_step": n,
            "sharding": "independent stream per GPU, no collectives",
            "host_api": "blocking adsb_demod_iq_device per step" if args.sync else
                        f"adsb_submit_iq_device / adsb_collect, {r['depth']} passes in flight",
            "kernels": "k_scan_fast (mag + sign planes + preamble + gates + trial syndromes) -> k_match -> "
                       "k_order_prefix -> k_records (bucket sort) -> host replay, or k_score -> k_emit on dense streams",
            "library": library,
            "runtime_env": {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")},
            "host_affinity": env.affinity_summary(),
            "clock_ramp": {**r["ramp"], "what": "untimed passes of this workload before the W warm-up steps, until the "
                           "GPU holds its clocks under the load (bench.py: clock_ramp); ms_per_step_cold = the first 20 "
                           "of them, i.e. what a run without the ramp reports"},
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "frac_of_achievable": round(achieved / HBM_ACHIEVABLE_GBS, 4),
            "frac_of_achievable_is": f"achieved / {HBM_ACHIEVABLE_GBS:.0f} GB/s, the guide's measured float4-copy bandwidth (79 % of the spec peak)",
            "traffic": None,
            "kernel": "k_scan_fast",
            "kernel_avg_ms": round(kernel_ms, 4),
            "kernel_avg_ms_is": "a launch's own start -> stop with blocking calls (launches do not overlap), HIP events "
                                "stamped by the launch on its stream" + ("" if args.sync else
                                ": the same steps once more after the timed region, untimed") +
                                "; what rocprofv3 --kernel-trace --stats shows for `bench.py --sync` "
                                "(profiles/*_sync_kernel_stats.csv)" if alone_ms else
                                "device time per launch: union of the overlapping launches' intervals / launches",
            "algorithmic_bytes_per_launch": algo,
            "launches_overlap_in_timed_region": not args.sync,
            "sustained": {
                "is": "what the pipelined stream of the timed region sustains (consecutive launches overlap on two streams)",
                "achieved_over_steps": round(algo * steps / r["elapsed"] / 1e9, 1),
                "frac_over_steps": round(algo * steps / r["elapsed"] / 1e9 / HBM_PEAK_GBS, 4),
                "device_ms_per_launch": round(excl_ms, 4),
                "device_ms_per_launch_is": "union of the overlapping launches' intervals / launches "
                                           "(adsb_stats.ms_scan_exclusive, HIP events stamped by the launches)",
                "frac_device": round(gbs(excl_ms) / HBM_PEAK_GBS, 4),
                "launch_own_duration_avg_ms": round(own_ms, 4),
            },
            "other_kernels_avg_ms": {"k_match_and_order": round(tot2["ms_match"] / steps, 4),
                                     "k_records": round(tot2["ms_records"] / steps, 4)},
            "device_chain_avg_ms": round(tot2["ms_total_device"] / steps, 4),
        },
        "device_stats_last_step": {k: r["stats"][k] for k in
                                   ("n_candidates", "n_ap_entries", "n_records", "n_messages", "retries")},
    }
    # Not measured in this run: counter passes cannot share a process with the timing.  Both come from
    # files committed under profiles/ for exactly this library build (PMC passes, tools/traffic.sh and
    # tools/sq_counters.py) and say so.
    tf = _profile_json("scan_hbm_traffic.json", library, args.chunks)
    if tf:
        result["roofline"]["traffic"] = tf.get("bytes_per_launch")
        result["roofline"]["traffic_source"] = ("read from profiles/scan_hbm_traffic.json (not measured in this run): "
                                                + str(tf.get("source")))
    # What bounds the kernel in practice: the HBM roofline above is what the path is
    # priced against; the scan itself is bound by vector-instruction issue.
    result["roofline"]["bound_in_practice"] = "valu-issue"
    sq = _profile_json("scan_sq_counters.json", library, args.chunks)
    if sq and sq.get("valu_roofline"):
        v = sq["valu_roofline"]
        floor_ms = v["wave_insts"] * v["busy_clocks_per_inst"] / (1024 * 2.4e9) * 1e3
        result["roofline"]["valu_floor_ms"] = round(floor_ms, 4)
        result["roofline"]["valu_floor_is"] = ("wave instructions per launch x measured busy clocks per instruction / (1024 SIMDs x "
                                               "2.4 GHz): the shortest launch this instruction stream allows; frac of the HBM "
                                               f"peak at that floor: {round(algo / (floor_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 3)}")
        result["roofline"]["valu"] = {
            "wave_insts": v["wave_insts"], "cycles_per_inst": v["busy_clocks_per_inst"],
            "simd_cycles_available": v["simd_cycles_available"], "frac": v["frac"],
            "is": v.get("is"), "wave_cycle_split": v.get("wave_cycle_split"),
            "source": "read from profiles/scan_sq_counters.json (rocprofv3 --pmc passes over `bench.py --sync` of this "
                      "library build, tools/sq_counters.py; not measured in this run)"}
    # ... and the two floors of the kernel beside it: the memory floor of its access pattern (the kernel cut after P1: the
    # whole HBM read, none of the later stages) and where its vector instructions go, stage by stage -- committed
    # measurements of this library build (tools/experiments/sessions/session_ablate.sh, tools/stage_split.py), not taken in this run.
    split = _profile_json("scan_stage_split.json", library, args.chunks)
    if split:
        result["roofline"]["memory_floor_ms"] = split["memory_floor_ms"]
        result["roofline"]["memory_floor_is"] = (split["memory_floor_is"] + f"; frac of the HBM peak at that floor: "
                                                 f"{round(algo / (split['memory_floor_ms'] / 1e3) / 1e9 / HBM_PEAK_GBS, 3)}")
        result["roofline"]["valu_by_stage"] = {"per_launch": split["valu_wave_insts_per_launch"],
                                               "stages": [{k: st[k] for k in ("stage", "valu_wave_insts", "per_wave_tile", "share")}
                                                          for st in split["stages"]],
                                               "source": "read from profiles/scan_stage_split.json (not measured in this run): " + split["source"]}
    return result


# ------------------------------------------------------------------------------------------------
# BASELINE config 3: streaming ring
# ------------------------------------------------------------------------------------------------
def run_stream(env: Env, chunks: int, steps: int, warmup: int, min_seconds: float = 0.0, check=False):
    """Sustained rate with the IQ starting in pinned host memory.  A step = one ring slot of `chunks`
    buffers: adsb_ring_submit launches a pass that reads the slot in place (one or two buffers per slot) or
    queues the slot's H2D copy in front of its pass on that pass's scan stream, while the other slots' passes run.  No icao_flush between steps (the live loop of main.rs never flushes).
    The ring buffers are filled once, outside the timed region (an SDR driver would DMA into them)."""
    from dump1090_rs_amd import Context, synth
    from dump1090_rs_amd._lib import AdsbMsg

    n = chunks * CHUNK
    ctx = Context(device=env.local_rank, max_chunks=chunks)
    ctx.ring_create(n)
    cap = 1 << 18
    out = (AdsbMsg * cap)()
    ctx.icao_flush()
    slots = ctx.max_in_flight()   # 4, or 8 for a ring of a few buffers per slot (one launch per pass)
    # three passes in flight for large slots (the fourth slot is being filled); a ring of one-launch
    # passes keeps every slot in flight: four run side by side, the others are queued behind them
    depth = 3 if slots == 4 else slots
    host_copies = []
    for k in range(slots):  # fill every pinned slot (and warm up)
        buf = ctx.ring_acquire()
        buf[:] = synth.make_iq(n, n_bursts=max(1, 64 * chunks // 512), seed=synth.SEED_DEFAULT + 7 * env.rank + k)
        if check:
            host_copies.append(buf.copy())
        ctx.ring_submit(n)
        ctx.collect_raw(out, cap)
    parity = None
    if check:
        # the same bytes through the CPU oracle as one stream: every slot in turn, then slot 0 again
        from oracle import binding
        orc = binding.Oracle()
        orc.icao_flush()
        for hc in host_copies:
            orc.demod_iq(hc, cap=cap)
        # second time round, pipelined the way the timed loop is (slots of three buffers and more are copied in
        # front of their pass when another pass is in flight, the first one is read in place)
        wants = [orc.demod_iq(hc, cap=cap)[0] for hc in host_copies]
        gots, sub = [], 0
        for k in range(slots):
            if sub - len(gots) >= depth:
                gots.append(ctx.collect(cap=cap))
            ctx.ring_acquire()
            ctx.ring_submit(n)
            sub += 1
        while len(gots) < sub:
            gots.append(ctx.collect(cap=cap))
        parity = all(_same(g, w) for g, w in zip(gots, wants))
    for _ in range(warmup):
        ctx.ring_acquire()
        ctx.ring_submit(n)
        ctx.collect_raw(out, cap)
    # A ring of one-launch passes is timed without HIP-event timing (the launch's own start / stop events
    # cost the submitting thread ~3 us of a ~16 us pass); the kernel's duration for the roofline note comes
    # from a short profiled run of the same loop afterwards.  Large slots keep level 1 (its cost is noise).
    lean = slots != 4
    ctx.set_profiling(0 if lean else 1)

    def loop(min_steps, seconds, timed_scan):
        frames, scan_ms, done, i = 0, 0.0, 0, 0
        t0 = time.perf_counter()
        while True:
            ctx.ring_acquire_raw()
            ctx.ring_submit(n)
            i += 1
            if i - done >= depth:
                frames += ctx.collect_raw(out, cap)
                if timed_scan:
                    scan_ms += ctx.stats_raw().ms_scan
                done += 1
            if i >= min_steps and (time.perf_counter() - t0) >= seconds:
                break
        while done < i:
            frames += ctx.collect_raw(out, cap)
            if timed_scan:
                scan_ms += ctx.stats_raw().ms_scan
            done += 1
        return i, frames, scan_ms

    # (no collector pause inside a loop of 15 us passes: after the resident legs this process's heap is large
    # and a full collection takes milliseconds -- the resident timed region does the same)
    gc.collect()
    gc.disable()
    env.fence()
    t0 = time.perf_counter()
    i, frames, scan_ms = loop(steps, min_seconds, not lean)
    env.fence()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if lean:
        ctx.set_profiling(1)
        k, _, ms = loop(2000, 0.0, True)
        scan_ms = ms / k * i
    ctx.close()
    return {"n": n, "steps": i, "elapsed": elapsed, "frames": frames, "scan_ms": scan_ms, "parity": parity,
            "depth": depth, "slots": slots}


def stream_result(env: Env, args, s):
    from dump1090_rs_amd import _lib
    elapsed, frames = env.reduce(s["elapsed"], s["frames"])
    n, steps = s["n"], s["steps"]
    scan_avg_s = s["scan_ms"] / steps / 1e3
    achieved = BYTES_PER_SAMPLE * n / scan_avg_s / 1e9 if scan_avg_s > 0 else 0.0
    return {
        "metric": "IQ Msamples/s demodulated", "value": round(n * steps * env.world / elapsed / 1e6, 1),
        "unit": "Msamples/s", "n_gpus": env.world, "steps": steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPE, "data": "synthetic", "frames_per_s": round(frames / elapsed, 1),
        "config": {"workload": f"streaming ring: {n // CHUNK} x 131072-sample buffers = {n * 4 // (1 << 20)} MiB per "
                               "slot, host-resident IQ, pinned hipMemcpyAsync ring inside the timed region "
                               "(BASELINE config 3)",
                   "h2d_GBps": round(BYTES_PER_SAMPLE * n * steps / s["elapsed"] / 1e9, 2),
                   "library": _lib.lib().adsb_version().decode()},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "kernel": "k_scan_fast",
                     "kernel_avg_ms": round(s["scan_ms"] / steps, 4),
                     "algorithmic_bytes_per_launch": BYTES_PER_SAMPLE * n,
                     "note": "PCIe-fed: the scan kernel idles between transfers; value is the sustained end-to-end rate"},
    }


def config3_leg(env: Env, args):
    """BASELINE config 3 as written -- a ring of 512 KB chunks, one read and one demodulation at a time
    (the reference's call shape, dump1090_rs/src/main.rs:161-167) -- next to larger slots: the ring at
    1, 4, 16 and 64 buffers per slot, the headline size (64 = 32 MiB) for --stream-seconds (>= 10 s,
    SURVEY 8d), the others for a quarter of that; every size checked against one oracle stream."""
    sweep = []
    for chunks in (1, 4, 16, 64):
        secs = args.stream_seconds if chunks == 64 else max(1.0, args.stream_seconds / 4)
        s = run_stream(env, chunks, 50, 3, min_seconds=secs, check=True)
        sr = stream_result(env, args, s)
        sweep.append({"buffers_per_slot": chunks, "slot_MiB": chunks * CHUNK * 4 / (1 << 20), "in_flight": s["depth"],
                      "value": sr["value"],
                      "unit": "Msamples/s", "seconds": round(s["elapsed"], 2), "steps": s["steps"],
                      "ms_per_slot": sr["ms_per_step"], "h2d_GBps": sr["config"]["h2d_GBps"],
                      "frames_per_s": sr["frames_per_s"], "parity_checked": s["parity"]})
    head = sweep[-1]
    return {"workload": "streaming ring (adsb_ring_*: pinned host slots; a slot of up to 16 buffers is ONE launch, eight in "
                        "flight on four streams: it reads a slot of one or two buffers in place over the link, larger slots are "
                        "copied by the copy engine on the pass's own stream in front of it; slots above 16 buffers: the copy, "
                        "then three launches, three in flight), host-resident IQ, the transfer inside the timed region "
                        "(BASELINE config 3)",
            "value": head["value"], "unit": "Msamples/s", "seconds": head["seconds"], "steps": head["steps"],
            "ms_per_step": head["ms_per_slot"], "h2d_GBps": head["h2d_GBps"],
            "value_512KB_slots": sweep[0]["value"],
            "parity_checked": all(x["parity_checked"] for x in sweep),
            "slot_sweep": sweep,
            "note": "PCIe-inclusive (host-resident IQ): never the headline value.  One 512 KB buffer per slot is the "
                    "reference's own call shape (main.rs:161-167): one launch per 131072 samples, bound by what a kernel "
                    "reads over the link in place (~36-39 GB/s); from three buffers per slot on by the copy engine (~52 GB/s)"}


def live_leg(env: Env, args, passes: int = 480):
    """The live receiver's loop as the reference runs it (dump1090_rs/src/main.rs:154-167): one read of
    131072 samples, one demodulation, for ever, the ICAO filter NEVER flushed.  `passes` DISTINCT buffers of
    one synthetic stream (a pool of 40 aircraft that keep being heard, so the filter is warm after the first
    buffers and still learns now and then) through the pinned ring, every slot in flight, each pass ONE launch
    that matches its address/parity trials inline against the address bitmap and writes no list for a second
    kernel; a pass that was in flight beside one that taught the filter a new address is redone by the host
    (adsb_host_rematches).  The whole frame list is compared with ONE oracle stream over the same bytes."""
    from dump1090_rs_amd import Context, synth
    from dump1090_rs_amd._lib import AdsbMsg
    from oracle import binding
    n = passes * CHUNK
    iq = synth.make_iq(n, n_bursts=12 * passes, seed=synth.SEED_DEFAULT + 4040, n_icao=40, df11_every=5)
    orc = binding.Oracle()
    orc.icao_flush()
    want, _ = orc.demod_iq(iq, cap=1 << 18)
    ctx = Context(device=env.local_rank, max_chunks=1)
    ctx.ring_create(CHUNK)
    ctx.icao_flush()
    depth = ctx.max_in_flight()
    got, done = [], 0
    t0 = time.perf_counter()
    import ctypes as C
    src = iq.ctypes.data
    for b in range(passes):
        if b - done >= depth:
            got += [(done, m) for m in ctx.collect()]
            done += 1
        # (the host fills the slot -- an SDR driver would DMA into it --: one 512 KB memcpy, no array objects made)
        C.memmove(ctx.ring_acquire_raw(), src + 4 * CHUNK * b, 4 * CHUNK)
        ctx.ring_submit(CHUNK)
    while done < passes:
        got += [(done, m) for m in ctx.collect()]
        done += 1
    elapsed = time.perf_counter() - t0
    same = [(s, m.j, m.try_phase, m.score, m.msg, m.signal_level) for s, m in got] == \
           [(w["chunk"], w["j"], w["try_phase"], w["score"], w["msg"], w["signal_level"]) for w in want]
    rematches = int(ctx._L.adsb_host_rematches(ctx._h))
    ctx.close()
    compiled = live_leg_compiled(iq, want, passes)
    return {"compiled_host": compiled,
            "workload": f"live receiver loop (main.rs:154-167): {passes} distinct 131072-sample buffers of one stream through "
                        f"the pinned ring, one launch per buffer, {depth} in flight, no icao_flush; the host copies every "
                        "buffer into its slot (an SDR read would land there)",
            "passes": passes, "frames": len(want), "parity_checked": bool(same), "passes_redone": rematches,
            "value": round(n / elapsed / 1e6, 1), "unit": "Msamples/s",
            "note": "value: this loop driven from Python -- it includes the host's 512 KB memcpy into the slot per pass (what an SDR's "
                    "DMA would do) and a Python list of message objects per pass; compiled_host: the same loop in C "
                    "(tests/abi_host --live), the library as a compiled caller sees it; the ring's own rate with the slots already "
                    "filled is config3_streaming_ring.slot_sweep[0]"}


def live_leg_compiled(iq, want, passes):
    """The same loop from a COMPILED host (tests/abi_host --live: C over include/adsb_hip.h, built by __graft_entry__.build()):
    no interpreter between the calls, the frames written out and compared here with the same oracle stream.  None when
    the helper is not built."""
    import tempfile
    exe = ROOT / "tests" / "abi_host"
    if not exe.exists():
        return None
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
        src, out = os.path.join(tmp, "stream.bin"), os.path.join(tmp, "frames.out")
        iq.tofile(src)
        best = None
        for _ in range(3):   # (a process start each: its first passes warm the context up; the best of three loops)
            r = subprocess.run([str(exe), "--live", src, out], capture_output=True, text=True, timeout=300)
            if r.returncode != 0 or not r.stdout.startswith("live: "):
                return {"error": (r.stdout + r.stderr)[-300:]}
            secs = float(r.stdout.split(",")[2].split()[0])
            best = secs if best is None else min(best, secs)
        got = [ln.split() for ln in open(out).read().splitlines()]
    import struct
    same = [(int(g[0]), int(g[1]), int(g[2]), int(g[3]), g[4], g[5]) for g in got] == \
           [(w["chunk"], w["j"], w["try_phase"], w["score"], w["buffer"].hex(), struct.pack(">d", w["signal_level"]).hex()) for w in want]
    return {"value": round(passes * CHUNK / best / 1e6, 1), "unit": "Msamples/s", "seconds": round(best, 6), "frames": len(got),
            "parity_checked": bool(same),
            "is": "tests/abi_host --live: acquire a ring slot, memcpy 512 KB into it, submit, collect the oldest when all are out -- "
                  "in C, best of three process runs (each starts cold); every frame compared with the oracle's stream"}


# ------------------------------------------------------------------------------------------------
# BASELINE config 4: one capture cut into contiguous buffer ranges, one per GPU
# ------------------------------------------------------------------------------------------------
def run_shard(env: Env, args):
    """ONE capture of world x --chunks buffers; rank r owns buffers [r*chunks, (r+1)*chunks).  A step =
    icao_flush + adsb_shard_scan on every rank, a host-side exchange of the addresses the shards
    learned (a few KB over a gloo group: the one real exchange step, and it goes through the host --
    no RCCL collective on the data path), adsb_shard_finish, the trial records gathered on rank 0 and
    replayed once in global (buffer, j, try_phase) order: the reference's loop
    dump1090_rs/src/main.rs:161-167 over the whole capture.  The second half of step i (finish, gather,
    replay) runs on a worker thread while the main thread scans step i + 1 on a second context
    (sharding.ShardPipeline).  Checked: the merged frame list equals the single-stream result of rank 0
    demodulating the whole capture alone."""
    torch = env.torch
    from dump1090_rs_amd import Context, sharding, synth, _lib

    total_chunks = max(env.world, args.capture_chunks)
    ranges = [sharding.chunk_range(total_chunks, env.world, r) for r in range(env.world)]
    first, last = ranges[env.rank]
    n = (last - first) * CHUNK

    def shard_iq(r, device):
        k = ranges[r][1] - ranges[r][0]
        return synth.make_iq_torch(k * CHUNK, n_bursts=max(1, 64 * k // 512), seed=synth.SEED_DEFAULT + 31 * r, device=device)

    mine = shard_iq(env.rank, env.dev)
    torch.cuda.synchronize()
    # (each context on its own stream: the worker thread's finish must not wait for the other context's scan)
    max_chunks = max(b - a for a, b in ranges)
    ctxs = [Context(device=env.local_rank, max_chunks=max_chunks) for _ in range(2)]
    base = first

    def blocking_step():
        ctxs[0].icao_flush()
        return sharding.demod_sharded(ctxs[0], mine.data_ptr(), n, base, env.dist)

    # the two blocking phases, one step at a time (what round 2 timed): for comparison, and as the ramp
    ramp = clock_ramp(env, args, lambda first, count: [blocking_step() for _ in range(count)])
    env.fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        blocking_step()
    env.fence()
    blocking_ms = (time.perf_counter() - t0) / args.steps * 1e3

    # Overlapping the two halves of consecutive steps pays when there is an exchange to hide (N > 1:
    # 0.79 vs 0.87 ms with two gloo ranks on one GPU); at N = 1 the hand-over to the worker thread costs
    # more than it hides (0.37 vs 0.31 ms), so a single rank keeps the blocking form.
    overlapped = env.world > 1
    pipe = sharding.ShardPipeline(ctxs, env.dist)
    for _ in range(max(2, args.warmup)):
        pipe.submit(mine.data_ptr(), n, base)
    pipe.drain()
    env.fence()
    t0 = time.perf_counter()
    frames, results = 0, 0
    for _ in range(args.steps):
        out = pipe.submit(mine.data_ptr(), n, base) if overlapped else blocking_step()
        if out is not None:
            frames += len(out)
            results += 1
    for out in pipe.drain():
        if out is not None:
            frames += len(out)
            results += 1
    env.fence()
    elapsed = time.perf_counter() - t0
    per_rank_ms = env.gather(elapsed / args.steps * 1e3)
    pipe.submit(mine.data_ptr(), n, base)
    merged = pipe.drain()[-1]       # (checked below: the overlapped form's result, whichever form was timed)
    pipe.close()
    # the single-stream answer: rank 0 demodulates the whole capture alone (untimed)
    same, n_frames, parity = None, None, None
    if env.rank == 0:
        whole = torch.cat([mine] + [shard_iq(r, env.dev) for r in range(1, env.world)]) if env.world > 1 else mine
        with Context(device=env.local_rank, max_chunks=min(512, total_chunks)) as solo:
            solo.icao_flush()
            want = solo.demod_iq_device(whole.data_ptr(), total_chunks * CHUNK, cap=1 << 20)
        key = lambda m: (m.chunk, m.j, m.try_phase, m.score, m.msg, m.signal_level)
        same = [key(m) for m in merged] == [key(m) for m in want] and results == args.steps
        n_frames = len(want)
        # ... and the CPU oracle over the whole capture
        from oracle import binding
        orc = binding.Oracle()
        orc.icao_flush()
        with env.all_cores():
            oracle_want, _ = orc.demod_iq(whole.cpu().numpy(), cap=1 << 20, threads=_oracle_threads())
        parity = _same(merged, oracle_want)
        del whole
    elapsed, frames = env.reduce(elapsed, frames)
    result = {
        "metric": "IQ Msamples/s demodulated", "value": round(total_chunks * CHUNK * args.steps / elapsed / 1e6, 1),
        "unit": "Msamples/s", "n_gpus": env.world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "ms_per_step_two_blocking_phases": round(env.reduce(blocking_ms / 1e3, 0)[0] * 1e3, 4),
        "per_rank_ms_per_step": [round(x, 4) for x in per_rank_ms],
        "backend": env.backend if env.dist is not None else None, "world_size_seen": env.world,
        # the capture is the same whatever N is: total work fixed, per-GPU work = 1 / N of it
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "frames_per_s": round(frames / elapsed, 1),
        "config": {"workload": f"one capture of {total_chunks} buffers = {total_chunks * CHUNK * 4 // (1 << 20)} MiB "
                               f"cut into {env.world} contiguous ranges ({', '.join(str(b - a) for a, b in ranges)} buffers), "
                               "one per GPU, resident in HBM (BASELINE config 4); a step = icao_flush + shard scan on every rank + host-side "
                               "exchange of learned addresses + match + records gathered and replayed once on rank 0",
                   "per_gpu_samples_per_step": [(b - a) * CHUNK for a, b in ranges],
                   "sharding": "contiguous buffer ranges, the IQ never moves; learned addresses and trial records "
                               "exchanged as host tensors over gloo groups (no RCCL collective on the data path; "
                               "torch.distributed's default group only carries the barrier and the timing reduction)",
                   "host_api": "adsb_shard_scan / adsb_shard_finish / adsb_replay_records; " +
                               ("finish + gather + replay of step i on a worker thread beside the scan of step i + 1 "
                                "(two contexts, sharding.ShardPipeline)" if overlapped else
                                "two blocking phases per step (a single rank has no exchange to hide)"),
                   "library": _lib.lib().adsb_version().decode(),
                   "host_affinity": env.affinity_summary(),
                   "clock_ramp": ramp},
        "shard_merge_equals_single_stream": same, "parity_checked": parity, "parity_frames": n_frames,
    }
    ms_step = elapsed / args.steps * 1e3
    gbs = total_chunks * CHUNK * BYTES_PER_SAMPLE / (ms_step * 1e-3) / 1e9
    result["roofline"] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS * env.world, "unit": "GB/s",
                          "frac": round(gbs / (HBM_PEAK_GBS * env.world), 4), "traffic": None,
                          "is": f"{BYTES_PER_SAMPLE} B per sample x the capture's samples / the whole step (both shard phases, the exchange over "
                                f"gloo, the gathered replay), against {env.world} x 8 TB/s"}
    for c in ctxs:
        c.close()
    return result


def run_shard_single_process(env: Env, args):
    """BASELINE config 4 from ONE process: adsb_multi_* (include/adsb_hip.h) -- the capture's contiguous ranges
    resident on their devices, a context and a host thread per device inside the library, the learned addresses
    united in memory between the two shard phases, one ordered replay through one filter.  A step = icao_flush +
    the whole capture; steps are pipelined four deep (submit / collect) like the single-GPU headline, and timed
    blocking as well for the orchestration figures (adsb_multi_stats: host-clock spans of the two phases)."""
    torch = env.torch
    from dump1090_rs_amd import Context, sharding, synth, _lib
    from dump1090_rs_amd._lib import AdsbMsg
    from dump1090_rs_amd.multi import MultiContext
    import ctypes as C

    n_dev = torch.cuda.device_count()
    shards = args.contexts or n_dev
    devices = [k % n_dev for k in range(shards)]
    total_chunks = max(shards, args.capture_chunks)
    ranges = [sharding.chunk_range(total_chunks, shards, r) for r in range(shards)]
    per = max(b - a for a, b in ranges)

    def shard_iq(r):
        k = ranges[r][1] - ranges[r][0]
        return synth.make_iq_torch(k * CHUNK, n_bursts=max(1, 64 * k // 512), seed=synth.SEED_DEFAULT + 31 * r,
                                   device=torch.device("cuda", devices[r]))

    parts = [shard_iq(r) for r in range(shards)]
    for d in set(devices):
        torch.cuda.synchronize(d)
    multi = MultiContext(devices, per)
    ptrs = (C.c_void_p * shards)(*[C.c_void_p(t.data_ptr()) for t in parts])
    ns = (C.c_size_t * shards)(*[t.shape[0] for t in parts])
    cap = 1 << 20
    out = (AdsbMsg * cap)()
    depth = multi.max_in_flight()
    keys = ("ms_wall", "ms_phase1_max", "ms_phase2_max", "ms_phase1_span", "ms_phase2_span", "ms_exchange", "ms_replay")

    def run_steps(count, blocking, acc=None):
        frames, done = 0, 0
        for i in range(count):
            multi.icao_flush()
            multi.submit_raw(ptrs, ns)
            if blocking or i - done >= depth - 1:
                frames += multi.collect_raw(out, cap)
                done += 1
                if acc is not None:
                    st = multi.stats()
                    for k in keys:
                        acc[k] += st[k]
        while done < count:
            frames += multi.collect_raw(out, cap)
            done += 1
        return frames

    gc.collect()
    gc.disable()
    ramp = clock_ramp(env, args, lambda first, count: run_steps(count, False))
    run_steps(max(2, args.warmup), False)
    env.fence()
    t0 = time.perf_counter()
    frames = run_steps(args.steps, False)
    env.fence()
    elapsed = time.perf_counter() - t0
    blocks = []
    for _ in range(max(0, args.blocks)):
        env.fence()
        tb = time.perf_counter()
        run_steps(args.steps, False)
        env.fence()
        blocks.append((time.perf_counter() - tb) / args.steps * 1e3)
    # one capture at a time: what a step costs from submit to its last record, and what of it is orchestration
    run_steps(5, True)
    acc = {k: 0.0 for k in keys}
    env.fence()
    t0 = time.perf_counter()
    run_steps(args.steps, True, acc)
    env.fence()
    blocking_ms = (time.perf_counter() - t0) / args.steps * 1e3
    gc.enable()
    orch = {k: round(v / args.steps, 4) for k, v in acc.items()}
    orch["ms_overhead"] = round(orch["ms_wall"] - orch["ms_phase1_span"] - orch["ms_phase2_span"], 4)
    orch["is"] = ("blocking steps, host clock, mean per step: ms_wall = submit -> the last device's records on the host; "
                  "ms_phase*_span = first device's issue of the phase -> last device's summary of it; ms_overhead = ms_wall "
                  "minus the two spans = hand-over to the device threads + the address union + hand-over of phase 2 + "
                  "noticing the end; ms_replay is the caller's ordered replay after that")
    # checked: the merged list of one more step == the single-stream answer of one context over the whole capture
    multi.icao_flush()
    merged = multi.demod_iq_device([t.data_ptr() for t in parts], [t.shape[0] for t in parts], cap=cap)
    whole = torch.cat([t.to(env.dev) for t in parts]) if shards > 1 else parts[0]
    with Context(device=env.local_rank, max_chunks=min(512, total_chunks)) as solo:
        solo.icao_flush()
        want = solo.demod_iq_device(whole.data_ptr(), total_chunks * CHUNK, cap=cap)
    key = lambda m: (m.chunk, m.j, m.try_phase, m.score, m.msg, m.signal_level)
    same = [key(m) for m in merged] == [key(m) for m in want]
    # ... and the CPU oracle over the whole capture (threads over buffers, one ordered replay)
    from oracle import binding
    orc = binding.Oracle()
    orc.icao_flush()
    with env.all_cores():
        oracle_want, _ = orc.demod_iq(whole.cpu().numpy(), cap=cap, threads=_oracle_threads())
    parity = _same(merged, oracle_want)
    del whole
    n = total_chunks * CHUNK
    ms_step = elapsed / args.steps * 1e3
    n_real = len(set(devices))
    gbs = n * BYTES_PER_SAMPLE / (ms_step * 1e-3) / 1e9
    result = {
        "metric": "IQ Msamples/s demodulated", "value": round(n * args.steps / elapsed / 1e6, 1), "unit": "Msamples/s",
        "n_gpus": len(set(devices)), "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "ms_per_step_blocking": round(blocking_ms, 4),
        "ms_per_step_blocks": None if not blocks else {"min": round(min(blocks), 4), "median": round(_median(blocks), 4),
                                                       "max": round(max(blocks), 4), "all": [round(x, 4) for x in blocks]},
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": DTYPE, "data": "synthetic",
        "frames_per_s": round(frames / elapsed, 1), "single_process": True, "shards": shards, "devices": devices,
        "orchestration": orch,
        "config": {"workload": f"one capture of {total_chunks} buffers = {n * 4 // (1 << 20)} MiB cut into {shards} contiguous "
                               f"ranges ({', '.join(str(b - a) for a, b in ranges)} buffers) resident on devices {devices} "
                               "(BASELINE config 4), ONE process: a step = adsb_multi_icao_flush + the capture through "
                               "adsb_multi_submit_iq_device / adsb_multi_collect",
                   "per_shard_samples_per_step": [(b - a) * CHUNK for a, b in ranges],
                   "sharding": "contiguous buffer ranges, the IQ never moves; one context and one host thread per device inside "
                               "libadsb_hip.so; learned addresses united in memory between the two shard phases; one ordered "
                               "replay through one filter on the caller's thread; no process group, no collective",
                   "host_api": f"adsb_multi_submit_iq_device / adsb_multi_collect, {depth} captures in flight",
                   "library": _lib.lib().adsb_version().decode(), "host_affinity": env.affinity_summary(), "clock_ramp": ramp},
        "shard_merge_equals_single_stream": bool(same), "parity_checked": bool(parity), "parity_frames": len(oracle_want),
        "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS * n_real, "unit": "GB/s",
                     "frac": round(gbs / (HBM_PEAK_GBS * n_real), 4), "traffic": None,
                     "is": f"{BYTES_PER_SAMPLE} B per sample x the capture's samples / the whole step, against {n_real} x 8 TB/s "
                           "(step-level: every kernel of both shard phases, the exchange and the replay are inside)"},
    }
    multi.close()
    del parts
    torch.cuda.empty_cache()
    if not args.no_also:
        result["busy_sky"] = busy_sky_leg(env, args, devices, ranges, per, total_chunks)
        if result["busy_sky"]["shard_merge_equals_single_stream"] is False:
            result["shard_merge_equals_single_stream"] = False
    return result


def busy_sky_leg(env: Env, args, devices, ranges, per, total_chunks):
    """The same capture size at config 5's density (5000 bursts per 512 buffers: ~180 frames a second of signal) through
    adsb_multi_*: the shards list their addresses while they scan, order their records on the device, and the collector
    scores a capture with several host threads at once -- checked against one context's single stream."""
    torch = env.torch
    from dump1090_rs_amd import Context, synth
    from dump1090_rs_amd._lib import AdsbMsg
    from dump1090_rs_amd.multi import MultiContext
    import ctypes as C
    shards = len(devices)
    parts = [synth.make_iq_torch((b - a) * CHUNK, n_bursts=max(1, 5000 * (b - a) // 512), seed=synth.SEED_DEFAULT + 77 * r + 5,
                                 device=torch.device("cuda", devices[r])) for r, (a, b) in enumerate(ranges)]
    for d in set(devices):
        torch.cuda.synchronize(d)
    cap = 1 << 20
    out = (AdsbMsg * cap)()
    with MultiContext(devices, per) as multi:
        ptrs = (C.c_void_p * shards)(*[C.c_void_p(t.data_ptr()) for t in parts])
        ns = (C.c_size_t * shards)(*[t.shape[0] for t in parts])
        depth = multi.max_in_flight()

        def run_steps(count):
            frames, done = 0, 0
            for i in range(count):
                multi.icao_flush()
                multi.submit_raw(ptrs, ns)
                if i - done >= depth - 1:
                    frames += multi.collect_raw(out, cap)
                    done += 1
            while done < count:
                frames += multi.collect_raw(out, cap)
                done += 1
            return frames

        run_steps(12)   # (the density of capture i is known when capture i + 1 starts; clocks are up from the sparse leg)
        env.fence()
        t0 = time.perf_counter()
        frames = run_steps(args.steps)
        env.fence()
        elapsed = time.perf_counter() - t0
        stats = multi.stats()
        multi.icao_flush()
        merged = multi.demod_iq_device([t.data_ptr() for t in parts], [t.shape[0] for t in parts], cap=cap)
        counters = multi.selftest_counters()
    whole = torch.cat([t.to(env.dev) for t in parts]) if shards > 1 else parts[0]
    with Context(device=env.local_rank, max_chunks=min(512, total_chunks)) as solo:
        solo.icao_flush()
        want = solo.demod_iq_device(whole.data_ptr(), total_chunks * CHUNK, cap=cap)
    key = lambda m: (m.chunk, m.j, m.try_phase, m.score, m.msg, m.signal_level)
    same = [key(m) for m in merged] == [key(m) for m in want]
    n = total_chunks * CHUNK
    return {"workload": f"the capture of {total_chunks} buffers with 5000 bursts per 512 buffers (config 5's density), {shards} shard(s), "
                        f"{depth} captures in flight, an icao_flush per capture",
            "value": round(n * args.steps / elapsed / 1e6, 1), "unit": "Msamples/s", "steps": args.steps,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "frames_per_step": frames // max(1, args.steps),
            "records_per_step": int(stats["n_records"]), "ms_replay_last_step": round(float(stats["ms_replay"]), 4),
            "device_ordered_shards": int(counters["device_ordered_shards"]), "parallel_scored_captures": int(counters["parallel_scored_captures"]),
            "shards_sorted_on_host": int(counters["shards_sorted_on_host"]),
            "shard_merge_equals_single_stream": bool(same), "parity_frames": len(want)}


def _oracle_threads() -> int:
    """threads for the checker over a whole capture: the CPUs this process may use, at most twice its cgroup's quota"""
    n = max(1, min(os.cpu_count() or 1, len(os.sched_getaffinity(0))))
    quota = _cpu_quota()
    return max(1, min(n, int(2 * quota + 0.5))) if quota else n


def config4_leg(env: Env, args, device_sets, steps: int):
    """BASELINE config 4 in the driver's line: ONE capture of --capture-chunks buffers (4096 = 2 GiB) through
    adsb_multi_* (one process, a context and a host thread per shard inside the library, one filter, one ordered
    replay -- the reference's loop dump1090_rs/src/main.rs:154-167 over N devices), for every entry of `device_sets`
    (lists of device indices: [0] = the whole capture on one context, [0] * 8 = eight shards on the one GPU, [0..N) = N
    real devices), a sparse sky (config 2's density) and a busy one (config 5's).  Per run: `steps` pipelined captures
    between two fences (an icao_flush each, four in flight), five blocking ones for the library's own host-clock spans,
    and one more flushed capture compared message by message with the CPU oracle over the WHOLE capture
    (orc_demod_iq_mt).  The capture is generated once per density on this rank's device; a shard on another device is
    a copy of its range."""
    torch = env.torch
    import ctypes as C
    from dump1090_rs_amd import sharding, synth, _lib
    from dump1090_rs_amd._lib import AdsbMsg
    from dump1090_rs_amd.multi import MultiContext
    from oracle import binding

    total_chunks = max(max(len(d) for d in device_sets), args.capture_chunks)
    n = total_chunks * CHUNK
    cap = 1 << 20
    out = (AdsbMsg * cap)()
    keys = ("ms_wall", "ms_phase1_max", "ms_phase2_max", "ms_phase1_span", "ms_phase2_span", "ms_exchange", "ms_replay")
    n_thr = _oracle_threads()
    runs = []
    all_same = True
    for sky, per_512 in (("sparse", 64), ("busy_sky", 5000)):
        whole = synth.make_iq_torch(n, n_bursts=max(1, per_512 * total_chunks // 512), seed=synth.SEED_DEFAULT + 4096 + per_512, device=env.dev)
        torch.cuda.synchronize()
        host = whole.cpu().numpy()
        orc = binding.Oracle()
        orc.icao_flush()
        with env.all_cores():
            timing = []
            want, _ = orc.demod_iq(host, cap=cap, threads=n_thr, timing=timing)
        del host
        for devices in device_sets:
            shards = len(devices)
            ranges = [sharding.chunk_range(total_chunks, shards, r) for r in range(shards)]
            per = max(b - a for a, b in ranges)
            parts = []
            for r, (a, b) in enumerate(ranges):
                t = whole[a * CHUNK:b * CHUNK]
                parts.append(t if devices[r] == env.local_rank else t.to(torch.device("cuda", devices[r])))
            for d in set(devices):
                torch.cuda.synchronize(d)
            with MultiContext(devices, per) as multi:
                ptrs = (C.c_void_p * shards)(*[C.c_void_p(t.data_ptr()) for t in parts])
                ns = (C.c_size_t * shards)(*[t.shape[0] for t in parts])
                depth = multi.max_in_flight()

                def run_steps(count, blocking=False, acc=None):
                    frames, done = 0, 0
                    for i in range(count):
                        multi.icao_flush()
                        multi.submit_raw(ptrs, ns)
                        if blocking or i - done >= depth - 1:
                            frames += multi.collect_raw(out, cap)
                            done += 1
                            if acc is not None:
                                st = multi.stats()
                                for k in keys:
                                    acc[k] += st[k]
                    while done < count:
                        frames += multi.collect_raw(out, cap)
                        done += 1
                    return frames

                run_steps(12)   # (clocks are up from the legs before; a stream's density is known one capture later)
                env.torch.cuda.synchronize()
                t0 = time.perf_counter()
                frames = run_steps(steps)
                env.torch.cuda.synchronize()
                elapsed = time.perf_counter() - t0
                acc = {k: 0.0 for k in keys}
                run_steps(5, True, acc)
                orch = {k: round(v / 5, 4) for k, v in acc.items()}
                orch["ms_overhead"] = round(orch["ms_wall"] - orch["ms_phase1_span"] - orch["ms_phase2_span"], 4)
                stats = multi.stats()
                multi.icao_flush()
                merged = multi.demod_iq_device([t.data_ptr() for t in parts], [t.shape[0] for t in parts], cap=cap)
                counters = multi.selftest_counters()
                wait = {_lib.ADSB_WAIT_SPIN: "spin", _lib.ADSB_WAIT_BLOCK: "block"}[multi.get_wait()]
            same = _same(merged, want)
            all_same = all_same and same
            ms = elapsed / steps * 1e3
            n_real = len(set(devices))
            gbs = n * BYTES_PER_SAMPLE / (ms * 1e-3) / 1e9
            runs.append({
                "sky": sky, "devices": devices, "shards": shards, "value": round(n * steps / elapsed / 1e6, 1), "unit": "Msamples/s",
                "steps": steps, "ms_per_step": round(ms, 4), "frames_per_step": frames // max(1, steps),
                "records_last_step": int(stats["n_records"]), "addresses_exchanged_last_step": int(stats["n_addrs_exchanged"]),
                "blocking_steps_host_clock": orch, "wait": wait,
                "device_ordered_shards": int(counters["device_ordered_shards"]), "device_scored_shards": int(counters["device_scored_shards"]),
                "parallel_scored_captures": int(counters["parallel_scored_captures"]),
                "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS * n_real, "unit": "GB/s",
                             "frac": round(gbs / (HBM_PEAK_GBS * n_real), 4),
                             "is": f"{BYTES_PER_SAMPLE} B per sample x the capture's samples / the whole step (every kernel of both shard "
                                   f"phases, the exchange, the replay), against {n_real} x 8 TB/s; the scan kernel's own figure is the headline's"},
                "parity_checked": bool(same), "parity_frames": len(want)})
            del parts
        del whole
        torch.cuda.empty_cache()
    return {"workload": f"one capture of {total_chunks} buffers = {n * 4 // (1 << 20)} MiB through adsb_multi_submit_iq_device / adsb_multi_collect "
                        "(ONE process, one filter, one ordered message list; BASELINE config 4), an icao_flush per capture, 4 captures in flight; "
                        "sparse: 64 bursts per 512 buffers, busy_sky: 5000",
            "runs": runs, "parity_checked": bool(all_same),
            "parity_is": f"every run's extra flushed capture against orc_demod_iq_mt over the whole capture ({n_thr} threads, "
                         f"{timing[0]:.2f} s for the last one)",
            "library": _lib.lib().adsb_version().decode()}


def one_process_n_devices_leg(env: Env, args):
    """--gpus N > 1, after the timed independent-stream region: every rank has closed its context and freed its buffers;
    rank 0 drives ONE adsb_multi over all N devices (config4_leg) while the others wait at a HOST-side barrier (a gloo
    group: an RCCL barrier would keep a spinning kernel on the very devices being measured).  Returns the leg on rank 0,
    None elsewhere."""
    torch = env.torch
    import datetime
    group = env.dist.new_group(backend="gloo", timeout=datetime.timedelta(minutes=15)) if env.dist is not None else None
    leg = None
    if env.rank == 0:
        try:
            n_vis = torch.cuda.device_count()
            devices = list(range(env.world)) if n_vis >= env.world else [k % n_vis for k in range(env.world)]
            # (the rank is pinned to the cores of ITS GPU's NUMA node; the library places a thread per device on that
            # device's node, within what the process may use: give it the whole mask back for this leg)
            with env.all_cores():
                leg = config4_leg(env, args, [devices], args.steps)
            leg["devices_visible_to_rank0"] = n_vis
            leg["is"] = (f"rank 0 alone, the other {env.world - 1} rank(s) idle at a host-side barrier: adsb_multi over devices {devices}")
        except Exception as e:   # the headline line must still come out; the failure is in it
            # (parity_checked stays None: nothing was compared -- a mismatch says False and fails the run, a leg that could
            # not run must not take the headline's exit status with it)
            leg = {"error": f"{type(e).__name__}: {e}", "parity_checked": None}
    if group is not None:
        env.dist.barrier(group=group)
    return leg


# ------------------------------------------------------------------------------------------------
# BASELINE config 1: the reference's `cargo bench` case
# ------------------------------------------------------------------------------------------------
def run_config1(env: Env):
    """icao_flush + to_mag + demodulate2400 on test_iq/test_1641427457780.iq (benches/demod_benchmark.rs:
    7-12), through the reference's two-call API shape and fused, with the CPU oracle (1 thread) beside it."""
    import numpy as np
    torch = env.torch
    from dump1090_rs_amd import Context
    from oracle import binding

    golden = json.loads((ROOT / "tests" / "golden" / "reference_frames.json").read_text())
    fx = golden["fixtures"][0]
    raw = np.fromfile(ROOT / "tests" / "golden" / fx["file"], dtype="<i2").reshape(-1, 2)
    iq = np.ascontiguousarray(raw[:, ::-1])  # file order is [im][re] (src/utils.rs:29-31)
    dev = torch.from_numpy(iq).to(env.dev)
    ctx = Context(env.local_rank, 1)

    def timeit(fn, seconds=0.6, warm=0.25):
        """mean over at least `seconds` of back-to-back calls, after `warm` seconds of them untimed (criterion
        warms up for 3 s and measures for 5; a burst of a few hundred calls after idling times the GPU at its
        idle clocks: 57 us where the sustained loop takes 30)"""
        t = time.perf_counter()
        while time.perf_counter() - t < warm:
            fn()
        torch.cuda.synchronize()
        t, reps = time.perf_counter(), 0
        while time.perf_counter() - t < seconds:
            for _ in range(50):
                fn()
            reps += 50
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps * 1e3

    def ref_api():
        ctx.icao_flush()
        return ctx.demodulate2400(ctx.to_mag(iq))

    def fused_host():
        ctx.icao_flush()
        return ctx.demod_iq(iq)

    def fused_dev():
        ctx.icao_flush()
        return ctx.demod_iq_device(dev.data_ptr(), len(iq))

    ok = all([m.buffer().hex() for m in f()] == fx["frames"] for f in (ref_api, fused_host, fused_dev))
    # the same buffer through the pinned ring, one pass at a time (the slot already holds the samples: an
    # SDR read lands there, main.rs:161): what a host that owns its buffers pays per call
    ctx.ring_create(len(iq))
    slots = ctx.max_in_flight()
    for _ in range(slots):
        ctx.ring_acquire()[:] = iq
        ctx.icao_flush()
        ctx.ring_submit(len(iq))
        ok = ok and [m.buffer().hex() for m in ctx.collect()] == fx["frames"]

    def ring_pinned():
        ctx.icao_flush()
        ctx.ring_acquire_raw()
        ctx.ring_submit(len(iq))
        return ctx.collect()

    # the two ABI calls alone, as a compiled caller makes them (no Python list of messages built per call)
    import ctypes as C
    from dump1090_rs_amd._lib import AdsbMsg
    raw_out, raw_n, L, h = (AdsbMsg * 4096)(), C.c_size_t(), ctx._L, ctx._h
    iq_ptr, dev_ptr = iq.ctypes.data, C.c_void_p(dev.data_ptr())

    def abi_host():
        L.adsb_icao_flush(h)
        L.adsb_demod_iq(h, iq_ptr, len(iq), raw_out, 4096, C.byref(raw_n))

    def abi_dev():
        L.adsb_icao_flush(h)
        L.adsb_demod_iq_device(h, dev_ptr, len(iq), raw_out, 4096, C.byref(raw_n))

    # the caller's own buffer registered once (adsb_host_register: main.rs reads the SDR into one Vec for ever)
    own = iq.copy()
    own_ptr = own.ctypes.data

    def abi_registered():
        L.adsb_icao_flush(h)
        L.adsb_demod_iq(h, own_ptr, len(own), raw_out, 4096, C.byref(raw_n))

    # the benchmark's own loop shape pipelined: icao_flush + one buffer per call (benches/demod_benchmark.rs:9-11), eight
    # calls in flight -- what a host that keeps the GPU fed pays per reference-shaped call.  Every pass starts from a
    # flushed filter, so every pass's frames are the capture's own (checked below on 64 of them).
    depth = ctx.max_in_flight()

    def pipelined(count):
        inflight, got = 0, 0
        for _ in range(count):
            L.adsb_icao_flush(h)
            L.adsb_submit_iq_device(h, dev_ptr, len(iq))
            inflight += 1
            if inflight >= depth:
                L.adsb_collect(h, raw_out, 4096, C.byref(raw_n))
                got += raw_n.value
                inflight -= 1
        while inflight:
            L.adsb_collect(h, raw_out, 4096, C.byref(raw_n))
            got += raw_n.value
            inflight -= 1
        return got

    t = time.perf_counter()
    while time.perf_counter() - t < 0.25:
        pipelined(200)
    torch.cuda.synchronize()
    t, reps, frames_seen = time.perf_counter(), 0, 0
    while time.perf_counter() - t < 0.6:
        frames_seen += pipelined(400)
        reps += 400
    torch.cuda.synchronize()
    ms_pipelined_flush_each = (time.perf_counter() - t) / reps * 1e3
    ok = ok and frames_seen == reps * len(fx["frames"])
    inflight, lists = 0, []
    for _ in range(64):
        ctx.icao_flush()
        ctx.submit_iq_device(dev.data_ptr(), len(iq))
        inflight += 1
        if inflight >= depth:
            lists.append(ctx.collect())
            inflight -= 1
    while inflight:
        lists.append(ctx.collect())
        inflight -= 1
    ok = ok and all([m.buffer().hex() for m in msgs] == fx["frames"] for msgs in lists)

    out = {"workload": "icao_flush + to_mag + demodulate2400 on test_1641427457780.iq, 131072 samples "
                       "(benches/demod_benchmark.rs:7-12; BASELINE config 1)",
           "ms_to_mag_plus_demodulate2400": round(timeit(ref_api), 4),
           "ms_fused_host_iq": round(timeit(fused_host), 4),
           "ms_fused_resident_iq": round(timeit(fused_dev), 4),
           "ms_ring_pinned_iq": round(timeit(ring_pinned), 4),
           "ms_fused_host_iq_c_abi": round(timeit(abi_host), 4),
           "ms_fused_resident_iq_c_abi": round(timeit(abi_dev), 4),
           "ms_fused_registered_host_iq_c_abi": None,
           "ms_pipelined_flush_each": round(ms_pipelined_flush_each, 4),
           "ms_pipelined_flush_each_is": f"adsb_icao_flush + adsb_submit_iq_device (resident capture) per call, {depth} calls in flight, "
                                         "adsb_collect in order; per call; every call's frame count checked, 64 calls' frames compared",
           "timing": "mean of back-to-back calls over >= 0.6 s after 0.25 s of warm-up, each through the Python mirror "
                     "of the reference's API (dump1090_rs_amd.Context: a list of message objects built per call); "
                     "*_c_abi: adsb_icao_flush + the one ABI call, as a compiled caller makes them.  A call of one "
                     "buffer is ONE launch (k_scan_fast<FUSED>), host IQ is copied once into pinned memory and read in "
                     "place while the copy is still going",
           "frames": len(fx["frames"]), "parity_checked": bool(ok),
           "published_reference_ms": PUBLISHED_CONFIG1_MS,
           "published_reference_note": "README.md:107, Intel i7-7700K, 1 thread, the Rust binary (other hardware)"}
    ctx.host_register(own)
    ctx.icao_flush()
    out["parity_checked"] = bool(out["parity_checked"] and [m.buffer().hex() for m in ctx.demod_iq(own)] == fx["frames"])
    out["ms_fused_registered_host_iq_c_abi"] = round(timeit(abi_registered), 4)
    ctx.host_unregister(own)
    orc = binding.Oracle()
    reps, t = 30, time.perf_counter()
    for _ in range(reps):
        orc.icao_flush()
        data, k = orc.to_mag(iq)
        orc.demodulate2400(data, k)
    out["cpu_port_1_thread_ms"] = round((time.perf_counter() - t) / reps * 1e3, 4)
    out["note"] = "one buffer cannot fill the chip (17 tiles for 1024 workgroup slots): these are latencies"
    ctx.close()
    return out


def dry_run(args) -> int:
    """What the CPU tests run: the launch + rendezvous path of --gpus N with no device anywhere."""
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
        import torch
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
        total = int(t.item())
        dist.barrier()
        dist.destroy_process_group()
    else:
        total = 1
    if rank == 0:
        print(json.dumps({"metric": "IQ Msamples/s demodulated", "dry_run": True, "n_gpus": world,
                          "rank_sum": total, "value": None}), flush=True)
    return 0


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args))
    if args.dry_run:
        sys.exit(dry_run(args))
    env = Env(args)
    args.gpus = env.world

    if args.workload == "stream":
        chunks = args.chunks if args.chunks != 512 else 64
        s = run_stream(env, chunks, args.steps, args.warmup, min_seconds=args.stream_seconds, check=env.rank == 0)
        result = stream_result(env, args, s)
        result["parity_checked"] = s["parity"]
    elif args.workload == "live":
        s = run_stream(env, 1, args.steps, args.warmup, min_seconds=args.stream_seconds, check=env.rank == 0)
        result = stream_result(env, args, s)
        result["config"]["workload"] = ("live receiver loop: one 131072-sample buffer (512 KB) per pass through the pinned "
                                        "ring, one launch per pass read in place over the link, no icao_flush (main.rs:154-167)")
        result["parity_checked"] = s["parity"]
        if env.rank == 0:
            result["live_receiver"] = live_leg(env, args)
            result["parity_checked"] = bool(s["parity"]) and result["live_receiver"]["parity_checked"]
    elif args.workload == "shard" and args.single_process:
        result = run_shard_single_process(env, args)
    elif args.workload == "shard":
        result = run_shard(env, args)
    else:
        r = run_resident(env, args, args.workload, args.steps, args.warmup)
        result = resident_result(env, args, r, args.workload)
        if env.rank == 0 and not args.no_cpu_baseline:
            # N = 1: the CPU baseline beside the line; N > 1: the parity gate alone (rank 0's buffer 0
            # against the oracle), so that the line verifies itself wherever the driver runs it
            base, same, n_frames = parity_leg(env, r, args.chunks, baseline=env.world == 1)
            if base is not None:
                result["cpu_baseline"] = base
            result["parity_checked"] = bool(same)
            result["parity_frames"] = n_frames
            if not same:
                result["parity_error"] = "GPU frame list differs from the CPU oracle"
        r["ctx"].close()
        del r
        env.torch.cuda.empty_cache()
        only = set(x for x in args.also_only.split(",") if x)
        want_leg = lambda name: not only or name in only   # noqa: E731
        if env.dist is not None and not args.no_also and not args.sync and args.workload == "sparse":
            # the reference's shape -- one process, one filter -- over all N devices, by rank 0 (every rank takes part in the barrier)
            one = one_process_n_devices_leg(env, args)
            if env.rank == 0:
                result.setdefault("also", {})["config4_one_process_n_devices"] = one
        if env.rank == 0 and env.dist is None and not args.no_also and not args.sync and args.workload == "sparse":
            also = result.setdefault("also", {})

            def leg(key, fn):
                """one short leg; if it cannot run (out of memory, a missing tool ...) the line still comes out, with the
                reason in the leg's place and parity_checked None -- only a MISMATCH (False) fails the run"""
                try:
                    also[key] = fn()
                except Exception as e:
                    also[key] = {"error": f"{type(e).__name__}: {e}", "parity_checked": None}
                    env.torch.cuda.empty_cache()

            if want_leg("config1"):
                leg("config1_cargo_bench_case", lambda: run_config1(env))
            if want_leg("config3"):
                leg("config3_streaming_ring", lambda: config3_leg(env, args))
            if want_leg("live"):
                leg("live_receiver", lambda: live_leg(env, args))
            # (a dense pass is a longer chain -- scan, match, order, records, score, emit, replay -- so the fill and the
            # drain of the pipeline between the two fences weigh ~0.3 ms: 14 us per step in a block of 20, 1.4 in one
            # of 200.  This leg reports the steady state, over at least 200 steps, and the short block beside it.)
            def dense_leg():
                dargs = argparse.Namespace(**vars(args))
                dargs.steps = max(args.steps, 200)
                d = run_resident(env, dargs, "dense", dargs.steps, dargs.warmup, level2=False, short_block=args.steps)
                try:
                    dr = resident_result(env, dargs, d, "dense")
                    _, dsame, dframes = parity_leg(env, d, args.chunks, baseline=False) if not args.no_cpu_baseline else (None, None, None)
                    return {
                        "workload": dr["config"]["workload"], "value": dr["value"], "unit": "Msamples/s",
                        "steps": dargs.steps, "ms_per_step": dr["ms_per_step"], "ms_per_step_median": dr["ms_per_step_median"],
                        "ms_per_step_blocks": (dr["ms_per_step_blocks"] or {}).get("all"),
                        "ms_per_step_in_a_block_of": {"steps": args.steps, "ms_per_step": round(d["short_block_ms"], 4),
                                                      "is": "the same loop over only this many steps between the fences: the "
                                                            "pipeline's fill and drain (one pass's whole chain, ~0.3 ms) included once"},
                        "frames_per_step": dr["frames_per_step"], "kernel_avg_ms": dr["roofline"]["kernel_avg_ms"],
                        "device_ms_per_launch": dr["roofline"]["sustained"]["device_ms_per_launch"],
                        "n_records_last_step": dr["device_stats_last_step"]["n_records"],
                        "parity_checked": dsame, "parity_frames": dframes}
                finally:
                    d["ctx"].close()
                    del d
                    env.torch.cuda.empty_cache()

            if want_leg("config5"):
                leg("config5_dense", dense_leg)
            if want_leg("config4"):
                n_dev = env.torch.cuda.device_count()
                sets = [[env.local_rank], [env.local_rank] * 8] + ([list(range(n_dev))] if n_dev > 1 else [])
                leg("config4_sharded_capture", lambda: config4_leg(env, args, sets, args.steps))

    if env.rank == 0:
        print(json.dumps(result), flush=True)
    env.finish()
    if env.rank == 0:
        bad = result.get("parity_checked") is False or result.get("shard_merge_equals_single_stream") is False
        for entry in (result.get("also") or {}).values():
            bad = bad or entry.get("parity_checked") is False or (entry.get("compiled_host") or {}).get("parity_checked") is False
        if bad:
            sys.exit(3)


if __name__ == "__main__":
    main()
