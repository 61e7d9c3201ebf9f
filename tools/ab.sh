#!/bin/bash
cp dump1090_rs_amd/libadsb_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so' EXIT   # whatever ends the script, the production library is back
# A/B: run bench with alternative builds of the library (variants/lib_<tag>.so), alternating the
# builds AB_REPS times (numbers are only comparable within one GPU session).  AB_ARGS="--sync"
# makes launches not overlap, so that kernel_avg_ms is the kernel alone.
for rep in $(seq 1 ${AB_REPS:-3}); do
  for tag in "$@"; do
    cp variants/lib_$tag.so dump1090_rs_amd/libadsb_hip.so
    echo -n "$tag: "; timeout 120 python bench.py --steps ${AB_STEPS:-40} --warmup 3 --no-cpu-baseline --no-also $AB_ARGS 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*\|kernel_avg_ms\": [0-9.]*\|exclusive_avg_ms\": [0-9.]*" | tr '\n' ' '; echo
  done
done
