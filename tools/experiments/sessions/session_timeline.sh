#!/bin/bash
# Where a one-launch pass spends its time (ADSB_KERNEL_ACCT build: variants/lib_acct.so; ADSB_TIMELINE=3).
T=${TAG:-s}; mkdir -p gpurun_out; H=gpurun_out/${T}_timeline.log
cp dump1090_rs_amd/libadsb_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so' EXIT   # whatever ends the script, the production library is back
python tools/config1.py > gpurun_out/${T}_config1.log 2>&1
cp variants/lib_acct.so dump1090_rs_amd/libadsb_hip.so
for d in 1 8; do
  ADSB_TIMELINE=3 python tools/hosttime.py ring --chunks 1 --depth $d --profiling 0 --passes 4000 >> $H 2>&1
  ADSB_TIMELINE=3 python tools/hosttime.py resident --chunks 1 --depth $d --profiling 0 --passes 4000 >> $H 2>&1
done
cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so
grep -v "amdgpu.ids\|occupancy\|stream priorities" $H gpurun_out/${T}_config1.log
