#!/bin/bash
# Round 5: the parallel replay's stage times by pool size (tuning build, ADSB_POOL_WORKERS), one context, dense
cd ${GRAFT_REPO_ROOT:-.}
cp dump1090_rs_amd/libadsb_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so' EXIT
cp variants/lib_tune.so dump1090_rs_amd/libadsb_hip.so
for w in 1 2 3 6 10; do
  echo "== workers $w (+ the caller)"
  ADSB_POOL_WORKERS=$w ADSB_HOST_TIMES=1 python tools/multi_steps.py --contexts 1 --chunks 4096 --steps 12 --pipelined --bursts 5000 2>&1 | grep "parallel replay\|^{" | cut -c1-330
done
