"""Why is the one-buffer ring slower inside the full bench process than in a fresh one?  The same leg
(bench.run_stream at one buffer per slot, 1.5 s) at several points of the bench's own sequence."""
import sys, time, gc
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
args = bench.parse(["--steps", "20", "--warmup", "5"])
env = bench.Env(args)
def ring(label):
    s = bench.run_stream(env, 1, 50, 3, min_seconds=1.5, check=False)
    print(f"{label:55s} {s['n'] * s['steps'] / s['elapsed'] / 1e6:8.0f} Msamples/s  ({s['elapsed'] / s['steps'] * 1e6:.2f} us per pass)", flush=True)
ring("fresh process")
r = bench.run_resident(env, args, "sparse", args.steps, args.warmup)
ring("after the resident leg (its context still open)")
base, same, nf = bench.parity_leg(env, r, args.chunks, baseline=True)
ring("after the parity leg + CPU baseline")
r["ctx"].close(); del r; env.torch.cuda.empty_cache(); gc.collect()
ring("after closing the big context and freeing its buffers")
c1 = bench.run_config1(env)
ring("after the config-1 leg")
