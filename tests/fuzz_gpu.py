"""Randomised parity soak: the HIP path (through the C ABI) against the CPU oracle on seeded
random captures -- ragged lengths, mixed frame kinds (DF17, DF11, address/parity DF4/DF20),
overlapping and saturating bursts, adversarial periodic patches, reference and carry-over
semantics, and every entry point (blocking host / device, pipelined submit/collect, ring,
two-phase shards).  Test infrastructure (uses oracle/); run on the GPU box:

    python tests/fuzz_gpu.py --cases 200 --seed 1

Exits non-zero at the first mismatch and prints the case so it can be replayed.
"""
from __future__ import annotations

import argparse
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

CHUNK = 131072
PERIODS = [
    [18143, 6637, 18778, 14788, 3662, 8402, 2882, 16543],
    [17914, 17500, 12559, 14370, 1698, 4482, 12099, 13742, 8850, 18749, 13607],
    [4962, 12056, 7113, 852, 14360, 11334, 17932, 13963, 124, 14783, 17800, 508, 13806, 1459],
]


def make_case(rng, synth, max_chunks=8):
    n_chunks = int(rng.integers(1, max_chunks + 1))
    n = int(n_chunks * CHUNK - (rng.integers(0, CHUNK - 400) if rng.random() < 0.6 else 0))
    n -= n % 4  # device-resident entry points want 16-byte multiples between cuts; keep it simple
    n = max(n, 400)
    seed = int(rng.integers(1, 1 << 30))
    iq = synth.noise_numpy(n, seed=seed)
    if rng.random() < 0.15:
        iq //= int(rng.integers(2, 40))          # quiet capture
    if rng.random() < 0.1:
        iq = (iq.astype(np.int32) * 12).clip(-32768, 32767).astype(np.int16)  # loud, saturating
    icaos = [int(x) for x in rng.integers(1, 1 << 24, size=int(rng.integers(1, 12)))]
    bursts = []
    for _ in range(int(rng.integers(0, 40 * n_chunks))):
        icao = icaos[int(rng.integers(0, len(icaos)))]
        kind = rng.random()
        if kind < 0.5:
            frame = synth.df17_frame(icao, int(rng.integers(0, 1 << 56)))
        elif kind < 0.7:
            frame = synth.df11_frame(icao)
        elif kind < 0.85:   # short address/parity (DF 0, 4, 5)
            body = bytes([int(rng.choice([0x00, 0x20, 0x28])) | int(rng.integers(0, 8))]) + bytes(rng.integers(0, 256, 3).tolist())
            frame = body + (synth.crc24(body) ^ icao).to_bytes(3, "big")
        else:               # long address/parity (DF 16, 20, 21)
            body = bytes([int(rng.choice([0x80, 0xA0, 0xA8])) | int(rng.integers(0, 8))]) + bytes(rng.integers(0, 256, 10).tolist())
            frame = body + (synth.crc24(body) ^ icao).to_bytes(3, "big")
        tick = int(rng.integers(-200, 5 * n))
        if rng.random() < 0.2:  # hug a buffer edge
            tick = 5 * (CHUNK * int(rng.integers(0, n_chunks + 1)) - int(rng.integers(0, 330))) + int(rng.integers(0, 5))
        bursts.append(synth.Burst(tick, int(rng.integers(1500, 32000)), int(rng.integers(0, 16)), frame))
    synth.add_bursts(iq, bursts)
    for _ in range(int(rng.integers(0, 3))):
        if rng.random() < 0.5:
            amps = np.array(PERIODS[int(rng.integers(0, len(PERIODS)))], dtype=np.int16)
            a = int(rng.integers(0, max(1, n - 100)))
            b = min(n, a + int(rng.integers(50, 30000)))
            iq[a:b, 0] = np.tile(amps, (b - a) // len(amps) + 1)[: b - a]
            iq[a:b, 1] = 0
    return iq, seed


def key(m):
    return (m.chunk, m.j, m.try_phase, m.score, m.msglen, m.msg, m.signal_level)


def okey(w):
    return (w["chunk"], w["j"], w["try_phase"], w["score"], w["len"], w["msg"], w["signal_level"])


def dense_pipeline_case(rng, synth, Context, Oracle, torch, modes, case, seed):
    """A stream of passes that switches between the host's and the device's ordering / scoring:
    dense passes of 17-30 buffers (thousands of records), the odd sparse or small one, random
    icao_flush, up to three in flight -- every pass against the oracle fed the same sequence."""
    chunks = int(rng.integers(17, 31))
    n = chunks * CHUNK
    n_icao = int(rng.integers(3, 40))
    bufs, host = [], []
    for k in range(3):
        dense = rng.random() < 0.8
        h = synth.make_iq(n, n_bursts=int(chunks * (rng.integers(70, 110) if dense else rng.integers(0, 6))),
                          seed=int(rng.integers(1, 1 << 30)), n_icao=n_icao, df11_every=int(rng.integers(0, 5)))
        host.append(h)
        bufs.append(torch.from_numpy(h).cuda())
    torch.cuda.synchronize()

    def check(got, want, what):
        if [key(m) for m in got] != [okey(w) for w in want]:
            print(f"MISMATCH dense pipeline {case} (fuzz seed {seed}) at {what}: {len(got)} frames, {len(want)} expected")
            sys.exit(1)

    orc = Oracle()
    ctx = Context(0, 32)
    orc.icao_flush()
    ctx.icao_flush()
    pending = []   # expected outputs of the passes in flight
    # (sequence and expected outputs first: the submissions then follow each other at the host's pace)
    plan = []
    in_flight = 0
    for step in range(int(rng.integers(6, 14))):
        if rng.random() < 0.15 and in_flight == 0:     # a small blocking call in between (host-scored)
            k = int(rng.integers(0, 3))
            m = int(rng.integers(1, 4)) * CHUNK
            plan.append(("small", k, m, orc.demod_iq(host[k][:m])[0]))
            continue
        flush = rng.random() < 0.3
        if flush:
            orc.icao_flush()
        k = int(rng.integers(0, 3))
        pre = in_flight == 3
        if pre:
            in_flight -= 1
        in_flight += 1
        post = rng.random() < 0.4
        if post:
            in_flight -= 1
        plan.append(("pass", k, (flush, pre, post), orc.demod_iq(host[k], cap=1 << 18)[0]))
    for step, (what, k, arg, want) in enumerate(plan):
        if what == "small":
            check(ctx.demod_iq(host[k][:arg]), want, f"step {step} (small)")
            continue
        flush, pre, post = arg
        if flush:
            ctx.icao_flush()
        if pre:
            check(ctx.collect(cap=1 << 18), pending.pop(0), f"step {step}")
        ctx.submit_iq_device(bufs[k].data_ptr(), n)
        pending.append(want)
        if post:
            check(ctx.collect(cap=1 << 18), pending.pop(0), f"step {step}")
    while pending:
        check(ctx.collect(cap=1 << 18), pending.pop(0), "drain")
    step = len(plan)
    modes[("dense_pipeline", False)] = modes.get(("dense_pipeline", False), 0) + 1
    modes[("dense_pipeline:host_replays", False)] = modes.get(("dense_pipeline:host_replays", False), 0) + int(ctx._L.adsb_host_replays(ctx._h))
    modes[("dense_pipeline:passes", False)] = modes.get(("dense_pipeline:passes", False), 0) + int(step)
    ctx.close()


def mixed_pipeline_case(rng, synth, Context, Oracle, torch, modes, case, seed):
    """Cross-pass ordering: a stream of passes of very different sizes -- one to six buffers (their match
    and records run on their own scan stream) between passes of 17-48 (tail stream) -- over captures that
    share a handful of addresses, so that address/parity frames keep depending on what earlier passes
    learned; random icao_flush between submissions, up to four in flight, device-resident or through the
    ring (whose copy delays a pass's start).  Every pass against the oracle fed the same sequence."""
    # (enough addresses that passes keep learning new ones, and flushes often enough that "new" recurs)
    icaos = [int(x) for x in rng.integers(1, 1 << 24, size=int(rng.integers(4, 40)))]
    p_flush = float(rng.choice([0.15, 0.4, 0.6]))

    def capture(chunks):
        n = chunks * CHUNK - (int(rng.integers(0, 3000)) // 4 * 4 if rng.random() < 0.5 else 0)
        iq = synth.noise_numpy(n, seed=int(rng.integers(1, 1 << 30)))
        bursts = []
        for _ in range(int(rng.integers(2, 14)) * chunks // 2 + 2):
            icao = icaos[int(rng.integers(0, len(icaos)))]
            kind = rng.random()
            if kind < 0.35:
                frame = synth.df17_frame(icao, int(rng.integers(0, 1 << 56)))
            elif kind < 0.45:
                frame = synth.df11_frame(icao)
            elif kind < 0.8:
                body = bytes([int(rng.choice([0x00, 0x20, 0x28])) | int(rng.integers(0, 8))]) + bytes(rng.integers(0, 256, 3).tolist())
                frame = body + (synth.crc24(body) ^ icao).to_bytes(3, "big")
            else:
                body = bytes([int(rng.choice([0x80, 0xA0, 0xA8])) | int(rng.integers(0, 8))]) + bytes(rng.integers(0, 256, 10).tolist())
                frame = body + (synth.crc24(body) ^ icao).to_bytes(3, "big")
            # (late in the capture as often as early: what the NEXT pass needs is learned at the end)
            tick = int(5 * n * (1 - rng.random() ** 2)) if rng.random() < 0.5 else int(rng.integers(0, 5 * n))
            bursts.append(synth.Burst(min(tick, 5 * (n - 400)), int(rng.integers(6000, 30000)), int(rng.integers(0, 16)), frame))
        synth.add_bursts(iq, bursts)
        return iq

    sizes = [int(rng.integers(1, 7)) if rng.random() < 0.6 else int(rng.integers(17, 49)) for _ in range(4)]
    host = [capture(c) for c in sizes]
    # some of the large ones dense (thousands of trial records): the context then moves between host- and
    # device-side ordering / scoring as the stream's density changes, with passes of both kinds in flight
    for k, c in enumerate(sizes):
        if c >= 17 and rng.random() < 0.5:
            host[k] = synth.make_iq(c * CHUNK, n_bursts=int(c * rng.integers(70, 110)), seed=int(rng.integers(1, 1 << 30)),
                                    n_icao=int(rng.integers(3, 40)), df11_every=int(rng.integers(0, 5)))
    dev = [torch.from_numpy(h).cuda() for h in host]
    torch.cuda.synchronize()
    ring_cap = int(rng.choice([6, 24])) * CHUNK
    ctx = Context(0, 48)
    ctx.ring_create(ring_cap)
    orc = Oracle()
    orc.icao_flush()
    ctx.icao_flush()
    depth = int(rng.integers(2, 5))
    pending = []

    def check(got, want, what):
        if [key(m) for m in got] != [okey(w) for w in want]:
            print(f"MISMATCH mixed pipeline {case} (fuzz seed {seed}) at {what}: sizes {sizes}, {len(got)} frames, {len(want)} expected")
            for x, y in zip([okey(w) for w in want], [key(m) for m in got]):
                if x != y:
                    print(" first difference:", x, y)
                    break
            sys.exit(1)

    # the whole sequence is drawn and its expected output computed first, so that the submissions below
    # follow each other as fast as the host can issue them (the oracle takes milliseconds per pass)
    steps = int(rng.integers(6, 16))
    plan = []
    for step in range(steps):
        flush = rng.random() < p_flush
        k = int(rng.integers(0, len(host)))
        ring = len(host[k]) <= ring_cap and rng.random() < 0.4
        early = rng.random() < 0.3
        if flush:
            orc.icao_flush()
        plan.append((flush, k, ring, early, orc.demod_iq(host[k], cap=1 << 18)[0]))
    for step, (flush, k, ring, early, want) in enumerate(plan):
        if flush:
            ctx.icao_flush()
        if len(pending) == depth:
            check(ctx.collect(cap=1 << 18), pending.pop(0), f"step {step}")
        n = len(host[k])
        if ring:
            buf = ctx.ring_acquire()
            buf[:n] = host[k]
            ctx.ring_submit(n)
        else:
            ctx.submit_iq_device(dev[k].data_ptr(), n)
        pending.append(want)
        if early:
            check(ctx.collect(cap=1 << 18), pending.pop(0), f"step {step}")
    while pending:
        check(ctx.collect(cap=1 << 18), pending.pop(0), "drain")
    modes[("mixed_pipeline", False)] = modes.get(("mixed_pipeline", False), 0) + 1
    modes[("mixed_pipeline:passes", False)] = modes.get(("mixed_pipeline:passes", False), 0) + steps
    ctx.close()


def multi_case(rng, synth, MultiContext, Oracle, torch, modes, case, seed, faults=True):
    """adsb_multi_*: a sequence of captures of random length over a random number of contexts on the one GPU (the
    devices wherever there are several), random icao_flush, the host form and resident shards, blocking and up to four
    captures in flight -- every capture against ONE oracle stream fed the same sequence."""
    n_dev = torch.cuda.device_count()
    k = int(rng.choice([1, 2, 3, 5, 8]))
    devices = [int(rng.integers(0, n_dev)) if n_dev > 1 else 0 for _ in range(k)]
    per = int(rng.integers(1, 7))
    big = rng.random() < 0.3
    if big:   # contexts of more than 16 buffers: full bitmaps -- shards that list their fresh addresses as they scan, and hand
        #       their records over in replay order once a capture has been dense (up to 11 bursts per buffer here)
        k = int(rng.choice([1, 2, 3]))
        devices = [int(rng.integers(0, n_dev)) if n_dev > 1 else 0 for _ in range(k)]
        per = int(rng.integers(17, 21))
    icaos = [int(x) for x in rng.integers(1, 1 << 24, size=int(rng.integers(2, 24)))]

    def capture():
        chunks = int(rng.integers(0, k * per + 1)) if rng.random() < 0.9 else int(rng.integers(k * per + 1, 2 * k * per + 2))
        n = max(0, chunks * CHUNK - (int(rng.integers(0, CHUNK - 400)) // 4 * 4 if rng.random() < 0.6 and chunks else 0))
        iq = synth.noise_numpy(n, seed=int(rng.integers(1, 1 << 30)))
        bursts = []
        for _ in range(int(rng.integers(0, 12)) * max(1, chunks) if n > 4000 else 0):
            icao = icaos[int(rng.integers(0, len(icaos)))]
            kind = rng.random()
            if kind < 0.35:
                frame = synth.df17_frame(icao, int(rng.integers(0, 1 << 56)))
            elif kind < 0.45:
                frame = synth.df11_frame(icao)
            elif kind < 0.8:
                body = bytes([int(rng.choice([0x00, 0x20, 0x28])) | int(rng.integers(0, 8))]) + bytes(rng.integers(0, 256, 3).tolist())
                frame = body + (synth.crc24(body) ^ icao).to_bytes(3, "big")
            else:
                body = bytes([int(rng.choice([0x80, 0xA0, 0xA8])) | int(rng.integers(0, 8))]) + bytes(rng.integers(0, 256, 10).tolist())
                frame = body + (synth.crc24(body) ^ icao).to_bytes(3, "big")
            tick = int(rng.integers(2000, max(2001, 5 * (n - 400))))
            if rng.random() < 0.2:   # hug a buffer edge = often a shard boundary
                tick = max(2000, 5 * (CHUNK * int(rng.integers(1, max(2, chunks))) - int(rng.integers(0, 330))) + int(rng.integers(0, 5)))
            bursts.append(synth.Burst(min(tick, max(2000, 5 * (n - 400))), int(rng.integers(6000, 30000)), int(rng.integers(0, 16)), frame))
        if n > 4000:
            synth.add_bursts(iq, bursts)
        return iq

    caps = [capture() for _ in range(int(rng.integers(2, 6)))]
    multi = MultiContext(devices, per)
    # half of the sequences: every capture scored by several host threads, whatever its size; some: a fresh-address list of
    # two entries, so that shards fall back to reading the addresses out of their records
    # ... and who scores a dense stream's shards: their devices (half), the host, or the devices with the result refused
    multi.selftest_tune(fresh_cap=2 if rng.random() < 0.25 else 0, parallel_min=1 if rng.random() < 0.5 else 0,
                        score_mode=int(rng.choice([0, 0, 1, 2])))
    from dump1090_rs_amd import _lib
    from dump1090_rs_amd._lib import AdsbError
    # (round 6's draws come from a stream of their own: the sequences older seeds stand for -- 27182 / 45 is a test -- stay what they were)
    frng = np.random.default_rng([int(seed) & 0x7FFFFFFF, int(case), 606])
    multi.set_wait(int(frng.choice([_lib.ADSB_WAIT_AUTO, _lib.ADSB_WAIT_SPIN, _lib.ADSB_WAIT_BLOCK])))
    orc = Oracle()
    steps = int(rng.integers(4, 12))
    plan = []
    for _ in range(steps):
        flush = rng.random() < 0.3
        c = int(rng.integers(0, len(caps)))
        fits = len(caps[c]) <= k * per * CHUNK
        form = "host" if (not fits or rng.random() < 0.3) else ("device" if rng.random() < 0.3 else "submit")
        if form == "submit" and len(caps[c]) and rng.random() < 0.4:   # the asynchronous HOST form, pinned or ordinary memory
            form = "submit_pinned" if rng.random() < 0.5 else "submit_host"
        plan.append((flush, c, form))
    # a third of the sequences: one shard of one capture fails (include/adsb_hip.h, "When a capture fails") -- that capture
    # returns the error, what is in flight behind it and what is submitted later ADSB_ERR_POISONED, the restart (icao_flush
    # with nothing in flight) starts a fresh oracle stream
    fault_step = int(frng.integers(0, steps)) if frng.random() < 0.33 and faults else -1
    fault_kind = int(frng.choice([_lib.ADSB_FAULT_PHASE1, _lib.ADSB_FAULT_PHASE2, _lib.ADSB_FAULT_RECORDS]))
    fault_shard = int(frng.integers(0, k))
    resident = {}
    for c, iq in enumerate(caps):
        if len(iq) > k * per * CHUNK:
            continue
        ts, ptrs, ns = [], [], []
        for dev, (a, n) in zip(devices, multi.shard_ranges(len(iq))):
            t = torch.from_numpy(np.ascontiguousarray(iq[a:a + n])).to(f"cuda:{dev}") if n else None
            ts.append(t)
            ptrs.append(t.data_ptr() if n else 0)
            ns.append(n)
        resident[c] = (ts, ptrs, ns)
    for d in set(devices):
        torch.cuda.synchronize(d)
    pending = []   # per capture in flight: ("ok", the oracle's list) | ("fail",) | ("poisoned",)
    state = {"broken": False, "fail_collected": False, "faults": 0, "restarts": 0, "poisoned_returns": 0}

    def fail(what):
        print(f"MISMATCH multi case {case} (fuzz seed {seed}) at {what}: devices {devices}, {per} buffers each, "
              f"captures {[len(x) for x in caps]}, fault at step {fault_step} kind {fault_kind} shard {fault_shard}")
        sys.exit(1)

    def check(got, want, what):
        if [key(m) for m in got] != want:
            print(f"{len(got)} frames, {len(want)} expected")
            for x, y in zip(want, [key(m) for m in got]):
                if x != y:
                    print(" first difference:", x, y)
                    break
            fail(what)

    def expect_error(call, poisoned, what):
        try:
            call()
        except AdsbError as e:
            if (e.status == _lib.ADSB_ERR_POISONED) != poisoned or e.status == 0:
                fail(f"{what}: status {e.status}, poisoned expected: {poisoned}")
            return
        fail(f"{what}: no error")

    def collect_one(what):
        entry = pending.pop(0)
        if entry[0] == "ok":
            check(multi.collect(cap=1 << 18), entry[1], what)
        elif entry[0] == "fail":
            expect_error(lambda: multi.collect(cap=1 << 18), False, what + " (the failing capture)")
            state["fail_collected"] = True
        else:
            expect_error(lambda: multi.collect(cap=1 << 18), True, what + " (behind the failing capture)")
            state["poisoned_returns"] += 1

    depth = int(rng.integers(1, 5))
    pinned = {}
    for step, (flush, c, form) in enumerate(plan):
        blocking = not form.startswith("submit")
        while pending and (blocking or len(pending) == depth):   # the blocking forms want nothing in flight
            collect_one(f"step {step} (collect)")
        if state["broken"] and not pending:
            multi.icao_flush()                                   # the restart
            orc.icao_flush()
            state["broken"] = state["fail_collected"] = False
            state["restarts"] += 1
        elif flush and not state["broken"]:
            multi.icao_flush()
            orc.icao_flush()
        # (a host capture longer than the contexts hold is cut into pieces = several captures: the fault is aimed at one)
        inject = step == fault_step and not state["broken"] and len(caps[c]) <= k * per * CHUNK
        if inject:
            multi.selftest_fail(0, fault_shard, fault_kind)
            state["faults"] += 1
        expect = "poisoned" if state["broken"] else ("fail" if inject else "ok")
        want = [okey(w) for w in orc.demod_iq(caps[c], cap=1 << 18)[0]] if expect == "ok" else None
        if form == "host":
            call = lambda: multi.demod_iq(caps[c], cap=1 << 18)
        elif form == "device":
            call = lambda: multi.demod_iq_device(resident[c][1], resident[c][2], cap=1 << 18)
        elif form == "submit_host":
            call = lambda: multi.submit_iq(caps[c])
        elif form == "submit_pinned":
            if c not in pinned:
                pinned[c] = multi.host_alloc(len(caps[c]))
                pinned[c][:] = caps[c]
            call = lambda: multi.submit_iq(pinned[c])
        else:
            call = lambda: multi.submit_iq_device(resident[c][1], resident[c][2])
        if blocking:   # (nothing in flight, and a broken handle was restarted above: "ok" or "fail")
            if expect == "ok":
                check(call(), want, f"step {step} ({form})")
            else:
                expect_error(call, False, f"step {step} ({form}, the failing capture)")
                state["broken"] = state["fail_collected"] = True
        elif expect == "poisoned" and state["fail_collected"]:
            expect_error(call, True, f"step {step} ({form}, a submission to a poisoned handle)")
            state["poisoned_returns"] += 1
        else:
            call()
            pending.append(("ok", want) if expect == "ok" else (expect,))
            if expect == "fail":
                state["broken"] = True
    while pending:
        collect_one("drain")
    for name in ("faults", "restarts", "poisoned_returns"):
        modes[("multi:" + name, False)] = modes.get(("multi:" + name, False), 0) + state[name]
    modes[("multi", False)] = modes.get(("multi", False), 0) + 1
    modes[("multi:captures", False)] = modes.get(("multi:captures", False), 0) + steps
    ctr = multi.selftest_counters()
    for name in ("device_ordered_shards", "fresh_list_fallbacks", "device_scored_shards", "scored_results_used", "scored_results_refused"):
        modes[("multi:" + name, False)] = modes.get(("multi:" + name, False), 0) + int(ctr[name])
    modes[("multi:parallel_replays", False)] = modes.get(("multi:parallel_replays", False), 0) + int(multi.selftest_counters()["parallel_scored_captures"])
    modes[("multi:host_submits", False)] = modes.get(("multi:host_submits", False), 0) + sum(f in ("submit_host", "submit_pinned") for _, _, f in plan)
    multi.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-chunks", type=int, default=8, help="captures of 1..N buffers (ring slots: min(2, N) buffers)")
    ap.add_argument("--only", type=int, default=-1, help="replay just this case of the seed")
    ap.add_argument("--api", default="", help="with --only: force this entry point")
    ap.add_argument("--ringcuts", action="store_true", help="with --only: cut every 2 buffers, as the ring does")
    ap.add_argument("--dense", type=int, default=8,
                    help="also run this many dense pipelines: passes of 17-30 buffers with thousands of trial records "
                         "each (ordered and scored on the device), random flushes, small and sparse passes between")
    ap.add_argument("--mixed", type=int, default=8,
                    help="also run this many mixed pipelines: small and large passes in flight together over "
                         "captures that share addresses, random flushes, device-resident and ring-fed")
    ap.add_argument("--multi", type=int, default=8,
                    help="also run this many adsb_multi_* sequences: captures over 1-8 contexts, flushes, host / resident, "
                         "blocking and pipelined, against one oracle stream")
    ap.add_argument("--no-multi-faults", action="store_true",
                    help="--multi sequences without injected shard failures (a third of them have one by default)")
    args = ap.parse_args()
    import torch
    from dump1090_rs_amd import Context, sharding, synth
    from dump1090_rs_amd.multi import MultiContext
    from dump1090_rs_amd.context import replay_records
    from oracle import binding
    from oracle.binding import demod_iq_carry

    rng = np.random.default_rng(args.seed)
    ctx = Context(0, args.max_chunks)
    ring_chunks = min(2, args.max_chunks)
    ctx.ring_create(ring_chunks * CHUNK)
    shard_ctx = [Context(0, args.max_chunks), Context(0, args.max_chunks)]
    # a ring of larger slots: three buffers and more per slot are copied in front of their pass (on its own
    # stream) when another pass is in flight, the first one of a burst is read in place (adsb_ring.cpp)
    big_chunks = max(3, min(5, args.max_chunks))
    ctx_big = Context(0, big_chunks)
    ctx_big.ring_create(big_chunks * CHUNK)
    t0 = time.time()
    modes = {}
    for case in range(args.cases):
        iq, seed = make_case(rng, synth, args.max_chunks)
        n = len(iq)
        carry_mode = rng.random() < 0.35
        api = str(rng.choice(["host", "device", "pipelined", "ring", "ring", "shards", "magbuf"]))
        if carry_mode and api in ("shards", "magbuf"):
            api = "device"
        ncuts = int(rng.integers(1, 4))
        cut_draw = rng.integers(1, n, size=ncuts - 1)
        if args.only >= 0 and case != args.only:
            continue
        if args.only >= 0 and args.api:
            api = args.api
        modes[(api, carry_mode)] = modes.get((api, carry_mode), 0) + 1
        # cut the stream into 1-3 calls (filter and carry persist across them)
        cuts = sorted(set([0, n] + [int(x) // 4 * 4 for x in cut_draw]))
        if args.ringcuts:
            cuts = list(range(0, n, ring_chunks * CHUNK)) + [n]
        orc = binding.Oracle()
        carry = np.zeros((326, 2), np.int16)
        wants = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            w = demod_iq_carry(orc, iq[a:b], carry, cap=1 << 20)[0] if carry_mode else orc.demod_iq(iq[a:b], cap=1 << 20)[0]
            wants.append([okey(x) for x in w])
        ctx.set_carry_over(carry_mode)
        ctx.icao_flush()
        gots = []
        dev = torch.from_numpy(iq).cuda() if api in ("device", "pipelined", "shards") else None
        if api == "host":
            gots = [[key(m) for m in ctx.demod_iq(iq[a:b], cap=1 << 20)] for a, b in zip(cuts[:-1], cuts[1:])]
        elif api == "device":
            gots = [[key(m) for m in ctx.demod_iq_device(dev.data_ptr() + 4 * a, b - a, cap=1 << 20)]
                    for a, b in zip(cuts[:-1], cuts[1:])]
        elif api == "pipelined":
            pend = 0
            for a, b in zip(cuts[:-1], cuts[1:]):
                if pend == 2:
                    gots.append([key(m) for m in ctx.collect(cap=1 << 20)])
                    pend -= 1
                ctx.submit_iq_device(dev.data_ptr() + 4 * a, b - a)
                pend += 1
            while pend:
                gots.append([key(m) for m in ctx.collect(cap=1 << 20)])
                pend -= 1
        elif api == "ring":
            # the ring takes at most ring_chunks buffers per slot: cut accordingly (oracle redone to match);
            # every other time the ring of larger slots, a random slot length (3 .. 5 buffers, ragged), 2 .. 4 deep
            rctx, per_slot, depth = ctx, ring_chunks * CHUNK, 2
            if rng.random() < 0.5:
                rctx, depth = ctx_big, int(rng.integers(2, 5))
                per_slot = int(rng.integers(2 * CHUNK + 4, big_chunks * CHUNK + 1)) // 4 * 4
                rctx.set_carry_over(carry_mode)
                rctx.icao_flush()
                modes[("ring:copied", carry_mode)] = modes.get(("ring:copied", carry_mode), 0) + 1
            cuts = list(range(0, n, per_slot)) + [n]
            orc = binding.Oracle()
            carry = np.zeros((326, 2), np.int16)
            wants = []
            for a, b in zip(cuts[:-1], cuts[1:]):
                w = demod_iq_carry(orc, iq[a:b], carry, cap=1 << 20)[0] if carry_mode else orc.demod_iq(iq[a:b], cap=1 << 20)[0]
                wants.append([okey(x) for x in w])
            pend = 0
            for a, b in zip(cuts[:-1], cuts[1:]):
                if pend == depth:
                    gots.append([key(m) for m in rctx.collect(cap=1 << 20)])
                    pend -= 1
                buf = rctx.ring_acquire()
                buf[: b - a] = iq[a:b]
                rctx.ring_submit(b - a)
                pend += 1
            while pend:
                gots.append([key(m) for m in rctx.collect(cap=1 << 20)])
                pend -= 1
        elif api == "magbuf":
            # the reference's two-call shape, one 131072-sample buffer at a time (filter persists)
            from dump1090_rs_amd import MagnitudeBuffer
            orc = binding.Oracle()
            wants, gots = [], []
            for a in range(0, n, CHUNK):
                part = iq[a:a + CHUNK]
                data, length = orc.to_mag(part)
                m = ctx.to_mag(part)
                if m.length != length or not np.array_equal(m.data, data):
                    print(f"MISMATCH case {case}: to_mag differs in buffer {a // CHUNK}")
                    sys.exit(1)
                wants.append([okey(x) for x in orc.demodulate2400(data, length, cap=1 << 18)[0]])
                gots.append([key(x) for x in ctx.demodulate2400(m, cap=1 << 18)])
        else:  # shards: the whole capture as one stream over two contexts
            wants = [[okey(x) for x in binding.Oracle().demod_iq(iq, cap=1 << 20)[0]]]
            spans = [sharding.sample_range(n, 2, r) for r in range(2)]
            for c in shard_ctx:
                c.icao_flush()
            learned = [c.shard_scan(dev.data_ptr() + 4 * a, b - a) for c, (a, b) in zip(shard_ctx, spans)]
            union = np.unique(np.concatenate(learned)) if learned else np.zeros(0, np.uint32)
            recs = [c.shard_finish(union) for c in shard_ctx]
            merged = sharding.merge_records(recs, [a // CHUNK for a, _ in spans])
            gots = [[key(m) for m in replay_records(merged, cap=1 << 20)]]
        if gots != wants:
            print(f"MISMATCH case {case} (fuzz seed {args.seed}, noise seed {seed}): api={api} carry={carry_mode} "
                  f"n={n} cuts={cuts} frames want {[len(w) for w in wants]} got {[len(g) for g in gots]}")
            for w, g in zip(wants, gots):
                for x, y in zip(w, g):
                    if x != y:
                        print(" first difference:", x, y)
                        break
            sys.exit(1)
    if args.only < 0:
        for k in range(args.dense):
            dense_pipeline_case(np.random.default_rng([args.seed, k]), synth, Context, binding.Oracle, torch, modes, k, args.seed)
        for k in range(args.mixed):
            mixed_pipeline_case(np.random.default_rng([args.seed, 1000003, k]), synth, Context, binding.Oracle, torch, modes, k, args.seed)
        for k in range(args.multi):
            multi_case(np.random.default_rng([args.seed, 2000003, k]), synth, MultiContext, binding.Oracle, torch, modes, k, args.seed, faults=not args.no_multi_faults)
    print(f"{args.cases} cases identical in {time.time() - t0:.1f} s; modes: "
          + ", ".join(f"{k[0]}{'+carry' if k[1] else ''}={v}" for k, v in sorted(modes.items())))


if __name__ == "__main__":
    main()
