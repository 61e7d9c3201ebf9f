#!/bin/bash
# Round 5 A/B of two builds of the scan (variants/lib_<a>.so, lib_<b>.so), sparse and dense, pipelined and blocking:
# 18 tiles of 7284 (9216 = exactly 9 each).  needs variants/lib_tile17.so and variants/lib_tile18.so
cd ${GRAFT_REPO_ROOT:-.}
cp dump1090_rs_amd/libadsb_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so' EXIT   # whatever ends the script, the production library is back
run() { echo -n "$1 $2 $3: "; cp variants/lib_$1.so dump1090_rs_amd/libadsb_hip.so; timeout 200 python bench.py --workload $2 $3 --steps 40 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('ms/step', d['ms_per_step'], 'median', d['ms_per_step_median'], 'blocks', d['ms_per_step_blocks']['all'][:3], 'kernel', d['roofline']['kernel_avg_ms'], 'device', d['roofline']['sustained']['device_ms_per_launch'])"; }
for rep in 1 2 3; do
  for w in sparse dense; do
    for t in static dyn; do run $t $w ""; run $t $w --sync; done
  done
done
