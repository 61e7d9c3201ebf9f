#!/bin/bash
# A/B: run bench with alternative builds of the library (gpurun_lib_<tag>.so)
for tag in "$@"; do
  cp gpurun_lib_$tag.so dump1090_rs_amd/libadsb_hip.so
  for rep in 1 2; do
    echo -n "$tag: "; timeout 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*\|kernel_avg_ms\": [0-9.]*" | tr '\n' ' '; echo
  done
done
