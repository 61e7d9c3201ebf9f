#!/bin/bash
# Round 5, stream pool, fifth A/B: the one-buffer ring in a FRESH process on the two shared high-priority streams only,
# against four normal-priority ones; hosttime's resident small passes on both.
cd ${GRAFT_REPO_ROOT:-.}
cp dump1090_rs_amd/libadsb_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so' EXIT   # whatever ends the script, the production library is back
cp variants/lib_tune.so dump1090_rs_amd/libadsb_hip.so
ring() { echo -n "ring $2 buffer/slot, $1: "; ch=$2; shift; shift; env "$@" timeout 120 python bench.py --workload stream --chunks $ch --stream-seconds 3 --steps 50 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(d['value'], 'Msamples/s', d['ms_per_step'], 'ms/slot, parity', d['parity_checked'])"; }
for rep in 1 2; do
  for ch in 1 4; do
  ring "two shared high" $ch ADSB_POOL_SMALL=0 ADSB_FUSED_STREAMS=2
  ring "four normal" $ch ADSB_POOL_SMALL=2
  ring "four high of its own (round 4)" $ch ADSB_POOL_SMALL=3
  done
done
