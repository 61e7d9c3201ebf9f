#!/bin/bash
# A/B of the per-device stream pool (round 5): the large sparse stream's pipelined step with the library before the pool,
# with the pool's variants (tuning build: ADSB_POOL=0 four scan + four low streams, 1 low streams created first,
# 2 two low streams, 3 two scan + two low streams) and with streams of the context's own in the old creation order
# (ADSB_STREAM_PRIO=0,2,2).  needs variants/lib_prepool.so and variants/lib_tune.so
cd ${GRAFT_REPO_ROOT:-.}
cp dump1090_rs_amd/libadsb_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so' EXIT   # whatever ends the script, the production library is back
run() { echo -n "$1: "; shift; env "$@" timeout 120 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('ms/step', d['ms_per_step'], 'median', d['ms_per_step_median'], 'blocks', d['ms_per_step_blocks']['all'][:3], 'kernel_alone', d['roofline']['kernel_avg_ms'])"; }
for rep in 1 2; do
  cp variants/lib_prepool.so dump1090_rs_amd/libadsb_hip.so; run prepool X=1
  cp variants/lib_tune.so dump1090_rs_amd/libadsb_hip.so
  for v in 0 1 2 3; do run "pool variant $v" ADSB_POOL=$v; done
  run "private streams, old order" ADSB_STREAM_PRIO=0,2,2
done
