#!/bin/bash
# Ring slots: the library's own rule (no ADSB_RING_COPY: read in place up to two buffers per slot or with nothing
# else in flight, else copied on the pass's own scan stream in front of it) / always in place where a pass is one
# launch (=0) / always copied (=2) / alternating (=3).  Needs the tuning build (tools/mkvariants.sh
# tune="-DADSB_TUNING").
cp dump1090_rs_amd/libadsb_hip.so /tmp/lib_prod.so
trap 'cp /tmp/lib_prod.so dump1090_rs_amd/libadsb_hip.so' EXIT   # whatever ends the script, the production library is back
cp variants/lib_tune.so dump1090_rs_amd/libadsb_hip.so
run() { # mode chunks depth
  echo -n "mode $1: "
  if [ $1 = rule ]; then python tools/hosttime.py ring --chunks $2 --passes $((16000 / $2 + 200)) --depth $3 2>&1 | grep "^ring"
  else ADSB_RING_COPY=$1 python tools/hosttime.py ring --chunks $2 --passes $((16000 / $2 + 200)) --depth $3 2>&1 | grep "^ring"; fi
}
for rep in 1 2; do
  for ch in 1 2 3; do for m in 0 2 3 rule; do run $m $ch 8; done; done
  for d in 1 2 8; do for ch in 4 16; do for m in 0 2 rule; do run $m $ch $d; done; done; done
done
cp /tmp/lib_prod.so dump1090_rs_amd/libadsb_hip.so
