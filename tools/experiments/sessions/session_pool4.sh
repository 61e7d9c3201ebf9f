#!/bin/bash
# Round 5, stream pool, fourth A/B (the bench's order: large context first, then the ring): touching every pool stream with
# kernels at creation; four LOW-priority scan streams for small contexts; the ring on the two shared high-priority streams only.
cd ${GRAFT_REPO_ROOT:-.}
cp dump1090_rs_amd/libadsb_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so' EXIT   # whatever ends the script, the production library is back
cp variants/lib_tune.so dump1090_rs_amd/libadsb_hip.so
for rep in 1 2; do
echo -n "normal x4, touched with kernels: "; ADSB_POOL_SMALL=2 ADSB_POOL_EAGER=2 timeout 300 python tools/ring_history_probe.py 2>/dev/null | tail -1
echo -n "low x4: "; ADSB_POOL_SMALL=4 timeout 300 python tools/ring_history_probe.py 2>/dev/null | tail -1
echo -n "two shared high only: "; ADSB_POOL_SMALL=0 ADSB_FUSED_STREAMS=2 timeout 300 python tools/ring_history_probe.py 2>/dev/null | tail -1
echo -n "normal x4, three used: "; ADSB_POOL_SMALL=2 ADSB_FUSED_STREAMS=3 timeout 300 python tools/ring_history_probe.py 2>/dev/null | tail -1
done
