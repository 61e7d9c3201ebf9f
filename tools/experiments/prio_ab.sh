#!/bin/bash
cp dump1090_rs_amd/libadsb_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so' EXIT   # whatever ends the script, the production library is back
for rep in 1 2 3; do
for t in base o5r2; do
  cp gpurun_lib_$t.so dump1090_rs_amd/libadsb_hip.so
  for pr in "0,1,2" "2,1,0" "2,0,1" "1,0,2"; do
    echo -n "$t prio $pr: "; ADSB_STREAM_PRIO=$pr timeout 120 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-also 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*\|kernel_avg_ms\": [0-9.]*" | tr '\n' ' '; echo
  done
done
done
