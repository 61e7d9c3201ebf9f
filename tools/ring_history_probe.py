"""Why is the one-buffer ring slower inside the full bench process than in a fresh one?  The bench's own
sequence (resident leg, parity + CPU baseline, config 1, config 3) with or without a small context that is
created and used once BEFORE the large one:   python tools/ring_history_probe.py [early]"""
import sys, gc
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
args = bench.parse(["--steps", "20", "--warmup", "5", "--stream-seconds", "4"])
env = bench.Env(args)
mode = next((a for a in sys.argv[1:] if a.startswith("early")), None)
if mode:
    # early: a small context with a ring, 8 passes; early0: created and destroyed, nothing run; early1: + ring_create;
    # early2: 8 pipelined passes over resident samples, no ring
    from dump1090_rs_amd import Context, synth
    import torch
    with Context(0, 1) as c:
        if mode in ("early", "early1"):
            c.ring_create(131072)
        if mode == "early":
            for k in range(8):
                c.ring_acquire()[:] = synth.noise_numpy(131072, seed=k)
                c.ring_submit(131072)
        if mode == "early2":
            dev = torch.from_numpy(synth.noise_numpy(131072, seed=3)).cuda()
            torch.cuda.synchronize()
            for k in range(8):
                c.submit_iq_device(dev.data_ptr(), 131072)
        while c.pending():
            c.collect()
r = bench.run_resident(env, args, "sparse", args.steps, args.warmup)
r_elapsed = r["elapsed"] / args.steps
base, same, nf = bench.parity_leg(env, r, args.chunks, baseline=True)
r["ctx"].close(); del r; env.torch.cuda.empty_cache(); gc.collect()
c1 = bench.run_config1(env)
leg = bench.config3_leg(env, args)
large_ms = round(r_elapsed * 1e3, 4)
print(mode or "the bench's order", "large", large_ms, [(x["buffers_per_slot"], x["value"]) for x in leg["slot_sweep"]],
      "config1", c1["ms_fused_host_iq"], c1["ms_fused_resident_iq"], flush=True)
