"""Collect-to-collect intervals of the dense workload's pipelined steps (bench.run_resident): where the mean's distance from the
median comes from.  usage: python tools/dense_intervals.py [sparse|dense] [steps]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
w = sys.argv[1] if len(sys.argv) > 1 else "dense"
steps = sys.argv[2] if len(sys.argv) > 2 else "400"
args = bench.parse(["--steps", steps, "--warmup", "5", "--blocks", "0", "--workload", w])
env = bench.Env(args)
r = bench.run_resident(env, args, w, args.steps, args.warmup, level2=False)
iv = sorted(x * 1e6 for x in r["intervals"])
n = len(iv)
print(f"{w}: {n} intervals, mean {sum(iv) / n:.1f} us, median {iv[n // 2]:.1f}, p90 {iv[int(.9 * n)]:.1f}, p99 {iv[int(.99 * n)]:.1f}, max {iv[-1]:.1f}; "
      f"elapsed / steps {r['elapsed'] / args.steps * 1e6:.1f} us")
seq = [x * 1e6 for x in r["intervals"]]
print("first 60 in order:", " ".join(f"{x:.0f}" for x in seq[:60]))
big = [(i, round(x)) for i, x in enumerate(seq) if x > 2 * iv[n // 2]]
print("intervals above twice the median (index, us):", big[:40])
st = r["ctx"].stats()
print("host replays", int(r["ctx"]._L.adsb_host_replays(r["ctx"]._h)), "sorts", int(r["ctx"]._L.adsb_host_sorts(r["ctx"]._h)), st)
