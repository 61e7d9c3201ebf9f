#!/bin/bash
# A/B: run bench with alternative builds of the library (gpurun_lib_<tag>.so), blocking calls so
# that launches do not overlap and kernel_avg_ms is the kernel alone; alternate the builds a few
# times (numbers are only comparable within one GPU session).
for rep in 1 2 3; do
  for tag in "$@"; do
    cp gpurun_lib_$tag.so dump1090_rs_amd/libadsb_hip.so
    echo -n "$tag: "; timeout 120 python bench.py --steps 40 --warmup 3 --no-cpu-baseline --sync 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*\|kernel_avg_ms\": [0-9.]*" | tr '\n' ' '; echo
  done
done
