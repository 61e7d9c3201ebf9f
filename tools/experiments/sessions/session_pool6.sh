#!/bin/bash
# Round 5, stream pool: WHAT about an early small context makes the ring fast later in the process?
cd ${GRAFT_REPO_ROOT:-.}
cp dump1090_rs_amd/libadsb_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so' EXIT   # whatever ends the script, the production library is back
cp variants/lib_tune.so dump1090_rs_amd/libadsb_hip.so
for m in early0 early1 early2 early ""; do
  echo -n "four normal, ${m:-no early context}: "; ADSB_POOL_SMALL=2 timeout 300 python tools/ring_history_probe.py $m 2>/dev/null | tail -1
done
