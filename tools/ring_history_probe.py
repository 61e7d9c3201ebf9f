"""Why is the one-buffer ring slower inside the full bench process than in a fresh one?  The bench's own
sequence (resident leg, parity + CPU baseline, config 1, config 3) with or without a small context that is
created and used once BEFORE the large one:   python tools/ring_history_probe.py [early]"""
import sys, gc
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
args = bench.parse(["--steps", "20", "--warmup", "5", "--stream-seconds", "4"])
env = bench.Env(args)
if "early" in sys.argv[1:]:
    from dump1090_rs_amd import Context, synth
    with Context(0, 1) as c:
        c.ring_create(131072)
        for k in range(8):
            c.ring_acquire()[:] = synth.noise_numpy(131072, seed=k)
            c.ring_submit(131072)
        while c.pending():
            c.collect()
r = bench.run_resident(env, args, "sparse", args.steps, args.warmup)
base, same, nf = bench.parity_leg(env, r, args.chunks, baseline=True)
r["ctx"].close(); del r; env.torch.cuda.empty_cache(); gc.collect()
c1 = bench.run_config1(env)
leg = bench.config3_leg(env, args)
print("early small context" if "early" in sys.argv[1:] else "the bench's order", [(x["buffers_per_slot"], x["value"]) for x in leg["slot_sweep"]],
      "config1", c1["ms_fused_host_iq"], c1["ms_fused_resident_iq"], flush=True)
