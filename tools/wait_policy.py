"""adsb_multi's wait policy (include/adsb_hip.h: adsb_multi_set_wait) measured: one 2 GiB capture over eight contexts on
the one GPU, pipelined four deep, with the device threads spinning, blocking and on AUTO -- milliseconds per capture and
host CPU-seconds per capture (user + system time of the whole process, resource.getrusage) -- for a sparse and a busy sky.
Run it under `taskset -c 0-3`, `taskset -c 0-15` and without (tools/experiments/sessions/session_r6_wait.sh):

    python tools/wait_policy.py [--chunks 4096] [--steps 60]
"""
import argparse
import ctypes as C
import os
import resource
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402

from dump1090_rs_amd import _lib, sharding, synth  # noqa: E402
from dump1090_rs_amd._lib import AdsbMsg  # noqa: E402
from dump1090_rs_amd.multi import MultiContext  # noqa: E402

CHUNK = 131072


def cpu_seconds():
    r = resource.getrusage(resource.RUSAGE_SELF)
    return r.ru_utime + r.ru_stime


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--shards", type=int, default=8)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    shards = args.shards
    ranges = [sharding.chunk_range(args.chunks, shards, r) for r in range(shards)]
    per = max(b - a for a, b in ranges)
    cap = 1 << 20
    out = (AdsbMsg * cap)()
    print(f"# cpus in the affinity mask: {len(os.sched_getaffinity(0))}; cpu.max: "
          f"{open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else 'n/a'}", flush=True)
    for sky, per_512 in (("sparse", 64), ("busy_sky", 5000)):
        whole = synth.make_iq_torch(args.chunks * CHUNK, n_bursts=max(1, per_512 * args.chunks // 512), seed=synth.SEED_DEFAULT + 99 + per_512, device=dev)
        torch.cuda.synchronize()
        parts = [whole[a * CHUNK:b * CHUNK] for a, b in ranges]
        ptrs = (C.c_void_p * shards)(*[C.c_void_p(t.data_ptr()) for t in parts])
        ns = (C.c_size_t * shards)(*[t.shape[0] for t in parts])
        frames_ref = None
        for mode, name in ((_lib.ADSB_WAIT_SPIN, "spin"), (_lib.ADSB_WAIT_BLOCK, "block"), (_lib.ADSB_WAIT_AUTO, "auto")):
            with MultiContext([0] * shards, per) as multi:
                multi.set_wait(mode)
                got = {_lib.ADSB_WAIT_SPIN: "spin", _lib.ADSB_WAIT_BLOCK: "block"}[multi.get_wait()]
                depth = multi.max_in_flight()

                def run(count):
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

                run(16)
                torch.cuda.synchronize()
                c0, t0 = cpu_seconds(), time.perf_counter()
                frames = run(args.steps)
                torch.cuda.synchronize()
                t1, c1 = time.perf_counter(), cpu_seconds()
                if frames_ref is None:
                    frames_ref = frames
                assert frames == frames_ref, (frames, frames_ref)
                # one capture at a time: submit -> the caller has its messages (what a blocking caller waits for)
                for _ in range(3):
                    multi.icao_flush()
                    multi.submit_raw(ptrs, ns)
                    multi.collect_raw(out, cap)
                torch.cuda.synchronize()
                b0 = time.perf_counter()
                for _ in range(20):
                    multi.icao_flush()
                    multi.submit_raw(ptrs, ns)
                    multi.collect_raw(out, cap)
                one_at_a_time = (time.perf_counter() - b0) / 20
                print(f"{sky:9s} {name:5s} (in effect: {got:5s})  {1e3 * (t1 - t0) / args.steps:8.4f} ms per capture   "
                      f"{1e3 * (c1 - c0) / args.steps:8.3f} CPU-ms per capture   {(c1 - c0) / (t1 - t0):5.2f} CPUs busy   "
                      f"{frames // args.steps} frames per capture   one at a time {1e3 * one_at_a_time:7.4f} ms", flush=True)
        del parts, whole
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
