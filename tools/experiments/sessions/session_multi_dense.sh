#!/bin/bash
# Round 5: where a dense capture's time goes through adsb_multi_* (tuning build: ADSB_HOST_TIMES=1 prints every device
# thread's host time per capture by stage), one context and eight on the one GPU, sparse / dense / four times denser
cd ${GRAFT_REPO_ROOT:-.}
cp dump1090_rs_amd/libadsb_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so' EXIT
cp variants/lib_tune.so dump1090_rs_amd/libadsb_hip.so
for n in 1 8; do for b in 64 5000 20000; do
  echo "== contexts $n, $b bursts per 512 buffers"
  ADSB_HOST_TIMES=1 python tools/multi_steps.py --contexts $n --chunks 4096 --steps 12 --pipelined --bursts $b 2>&1 | grep -v amdgpu.ids | grep "device thread [07]:\|parallel replay\|^{" | cut -c1-640
done; done
