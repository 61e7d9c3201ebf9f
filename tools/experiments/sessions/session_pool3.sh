#!/bin/bash
# Round 5, stream pool, third A/B: four normal-priority scan streams for contexts of a few buffers (variant 2), created
# when the first such context is made (ADSB_POOL_EAGER=0) or with the device's other streams at the first adsb_create (1);
# both kinds of context in one process in both orders, twice.   needs variants/lib_tune.so
cd ${GRAFT_REPO_ROOT:-.}
cp dump1090_rs_amd/libadsb_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so' EXIT   # whatever ends the script, the production library is back
cp variants/lib_tune.so dump1090_rs_amd/libadsb_hip.so
for rep in 1 2; do
for e in 2 0; do
  echo -n "eager $e, large then ring: "; ADSB_POOL_SMALL=2 ADSB_POOL_EAGER=$e timeout 300 python tools/ring_history_probe.py 2>/dev/null | tail -1
  echo -n "eager $e, ring then large: "; ADSB_POOL_SMALL=2 ADSB_POOL_EAGER=$e timeout 300 python tools/ring_history_probe.py early 2>/dev/null | tail -1
done
done
echo -n "eager 2 large alone: "; ADSB_POOL_SMALL=2 ADSB_POOL_EAGER=2 timeout 120 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('ms/step', d['ms_per_step'], 'median', d['ms_per_step_median'], 'blocks', d['ms_per_step_blocks']['all'][:3])"
