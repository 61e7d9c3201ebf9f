"""Host-side cost of a pass, call by call.

    python tools/hosttime.py ring [--chunks 1] [--passes 20000] [--depth 3]   # one ring slot per pass
    python tools/hosttime.py resident [--chunks 1] [--depth 3]                # the same over device-resident IQ
    python tools/hosttime.py flush                                            # flush / stats / profiling levels

`ring` drives the streaming ring the way bench.py's config-3 leg does (acquire, submit, collect with
`depth` passes in flight) and times the three ABI calls from the caller's side; with a library built
with ADSB_HIPCC_FLAGS=-DADSB_TUNING and ADSB_HOST_TIMES=1 the library adds its own table on stderr when
the context is destroyed: microseconds per pass in each HIP call (hipMemcpyAsync, event records and
waits, the launches, the wait for the pass, checksum, replay).  tools/mkvariants.sh builds the tuning variant.
"""
import argparse
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402
from dump1090_rs_amd import Context, synth  # noqa: E402
from dump1090_rs_amd._lib import AdsbMsg  # noqa: E402

CHUNK = 131072


def ring(args):
    n = args.chunks * CHUNK
    ctx = Context(0, args.chunks)
    ctx.ring_create(n)
    cap = 1 << 16
    out = (AdsbMsg * cap)()
    ctx.icao_flush()
    args.depth = min(args.depth, ctx.max_in_flight())
    for k in range(ctx.max_in_flight()):
        buf = ctx.ring_acquire()
        buf[:] = synth.make_iq(n, n_bursts=max(1, 64 * args.chunks // 512), seed=synth.SEED_DEFAULT + k)
        ctx.ring_submit(n)
        ctx.collect_raw(out, cap)
    ctx.set_profiling(args.profiling)
    t_acq = t_sub = t_col = 0.0
    inflight = frames = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.passes):
        a = time.perf_counter()
        ctx.ring_acquire_raw()
        b = time.perf_counter()
        ctx.ring_submit(n)
        c = time.perf_counter()
        t_acq += b - a
        t_sub += c - b
        inflight += 1
        if inflight >= args.depth:
            frames += ctx.collect_raw(out, cap)
            t_col += time.perf_counter() - c
            inflight -= 1
    while inflight:
        frames += ctx.collect_raw(out, cap)
        inflight -= 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    us = 1e6 / args.passes
    print(f"ring, {args.chunks} buffer(s) per slot, {args.depth} in flight, profiling {args.profiling}: "
          f"{dt * us:.2f} us per pass = {n * args.passes / dt / 1e6:.0f} Msamples/s "
          f"({4 * n * args.passes / dt / 1e9:.1f} GB/s), {frames} frames", flush=True)
    print(f"  caller's side: adsb_ring_acquire {t_acq * us:.2f} us, adsb_ring_submit {t_sub * us:.2f} us, "
          f"adsb_collect {t_col * us:.2f} us (ctypes call overhead included)", flush=True)
    ctx.close()   # (a tuning build with ADSB_HOST_TIMES=1 prints its table here)


def resident(args):
    """the same loop over device-resident IQ (adsb_submit_iq_device / adsb_collect): what a pass costs
    without the link"""
    n = args.chunks * CHUNK
    ctx = Context(0, args.chunks)
    cap = 1 << 16
    out = (AdsbMsg * cap)()
    bufs = [synth.make_iq_torch(n, n_bursts=max(1, 64 * args.chunks // 512), seed=synth.SEED_DEFAULT + k, device="cuda")
            for k in range(4)]
    torch.cuda.synchronize()
    ctx.icao_flush()
    for b in bufs:
        ctx.demod_iq_device_raw(b.data_ptr(), n, out, cap)
    ctx.set_profiling(args.profiling)
    for flush_each in (False, True):
        inflight = frames = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.passes):
            if flush_each:
                ctx.icao_flush()
            ctx.submit_iq_device(bufs[i % 4].data_ptr(), n)
            inflight += 1
            if inflight >= args.depth:
                frames += ctx.collect_raw(out, cap)
                inflight -= 1
        while inflight:
            frames += ctx.collect_raw(out, cap)
            inflight -= 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"resident, {args.chunks} buffer(s) per pass, {args.depth} in flight, profiling {args.profiling}, "
              f"{'icao_flush before every pass' if flush_each else 'no flush'}: {dt * 1e6 / args.passes:.2f} us per pass = "
              f"{n * args.passes / dt / 1e6:.0f} Msamples/s, {frames} frames", flush=True)
    ctx.close()


def flush(_args):
    n = 512 * CHUNK
    bufs = [synth.make_iq_torch(n, n_bursts=64, seed=synth.SEED_DEFAULT + b, device='cuda') for b in range(3)]
    torch.cuda.synchronize()
    ctx = Context(0, 512)
    cap = 1 << 20
    out = (AdsbMsg * cap)()

    def run(label, flush=True, stats=True, prof=True, steps=30):
        ctx.set_profiling(prof)
        for i in range(3):
            ctx.icao_flush()
            ctx.demod_iq_device_raw(bufs[i % 3].data_ptr(), n, out, cap)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            if flush:
                ctx.icao_flush()
            ctx.demod_iq_device_raw(bufs[i % 3].data_ptr(), n, out, cap)
            if stats:
                ctx.stats()
        torch.cuda.synchronize()
        print(f"{label:40s} {(time.perf_counter() - t0) / steps * 1e6:8.1f} us/step")

    run("flush+demod+stats, profiling on")
    run("flush+demod, profiling on", stats=False)
    run("flush+demod, profiling off", stats=False, prof=False)
    run("demod only, profiling off", flush=False, stats=False, prof=False)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["ring", "resident", "flush"])
    ap.add_argument("--chunks", type=int, default=1)
    ap.add_argument("--passes", type=int, default=20000)
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--profiling", type=int, default=1)
    a = ap.parse_args()
    {"ring": ring, "resident": resident, "flush": flush}[a.mode](a)
