#!/bin/bash
# Round 5, stream pool + one pinned block per context / per ring: both kinds of context in one process, the bench's
# order and with a small context created first; then the production library through the whole bench line.
cd ${GRAFT_REPO_ROOT:-.}
cp dump1090_rs_amd/libadsb_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so' EXIT   # whatever ends the script, the production library is back
cp variants/lib_tune.so dump1090_rs_amd/libadsb_hip.so
for rep in 1 2; do
for m in "" early0; do
  echo -n "four normal, one block, ${m:-no early context}: "; timeout 300 python tools/ring_history_probe.py $m 2>/dev/null | tail -1
done
done
