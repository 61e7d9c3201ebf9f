#!/bin/bash
# Round 6 session C: the wait policy once more with the one-capture-at-a-time latency beside it; `bench.py --gpus 2` over gloo
# on the one GPU (the N > 1 path: timed independent streams, then rank 0's adsb_multi over "both devices").
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6_c
mkdir -p $O
for cpus in 0-3 all; do
  echo "## affinity: $cpus" >> $O/wait.txt
  if [ $cpus = all ]; then timeout 600 python tools/wait_policy.py >> $O/wait.txt 2>&1; else timeout 600 taskset -c $cpus python tools/wait_policy.py >> $O/wait.txt 2>&1; fi
done
grep -v amdgpu.ids $O/wait.txt
ADSB_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 10 --capture-chunks 2048 > $O/bench_gloo2.json 2> $O/bench_gloo2.err; echo "gloo2 rc=$?"; tail -3 $O/bench_gloo2.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6_c/bench_gloo2.json"))
print(d["value"], d["ms_per_step"], d["n_gpus"], d.get("parity_checked"))
leg = d["also"]["config4_one_process_n_devices"]
print({k: v for k, v in leg.items() if k != "runs"})
for r in leg.get("runs", []):
    print(r["sky"], r["devices"], r["value"], r["ms_per_step"], r["parity_checked"], r["wait"], r["blocking_steps_host_clock"])
PY
