#!/bin/bash
# Round 5, stream pool, second A/B: (a) the large sparse stream with two high-priority streams and 2 / 4 low ones;
# (b) the one-buffer ring with its four scan streams as 0: two shared high + two more high, 1: two shared high + two of
# normal priority, 2: four of normal priority, 3: four high of its own; (c) both kinds in ONE process, large first
# (tools/ring_history_probe.py) -- the case that needed GPU_MAX_HW_QUEUES=8.   needs variants/lib_tune.so
cd ${GRAFT_REPO_ROOT:-.}
cp dump1090_rs_amd/libadsb_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so' EXIT   # whatever ends the script, the production library is back
cp variants/lib_tune.so dump1090_rs_amd/libadsb_hip.so
large() { echo -n "large, $1: "; shift; env "$@" timeout 120 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('ms/step', d['ms_per_step'], 'median', d['ms_per_step_median'], 'blocks', d['ms_per_step_blocks']['all'][:3])"; }
ring() { echo -n "ring 1 buffer/slot, $1: "; shift; env "$@" timeout 120 python bench.py --workload stream --chunks 1 --stream-seconds 3 --steps 50 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms/slot, parity', d['parity_checked'])"; }
for rep in 1 2; do
  large "2 low" ADSB_POOL_LOW=2; large "4 low" ADSB_POOL_LOW=4
  for v in 0 1 2 3; do ring "small variant $v" ADSB_POOL_SMALL=$v; done
done
for v in 1 2 0; do
  echo -n "one process, large then ring, small variant $v: "; ADSB_POOL_SMALL=$v timeout 300 python tools/ring_history_probe.py 2>/dev/null | tail -1
  echo -n "one process, ring first then large, small variant $v: "; ADSB_POOL_SMALL=$v timeout 300 python tools/ring_history_probe.py early 2>/dev/null | tail -1
done
