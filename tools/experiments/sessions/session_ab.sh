#!/bin/bash
# A/B of library variants (variants/lib_<tag>.so) on the headline and the dense workload, alternating.
T=${TAG:-s}; mkdir -p gpurun_out; O=gpurun_out/${T}_ab.log
cp dump1090_rs_amd/libadsb_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so' EXIT   # whatever ends the script, the production library is back
for rep in 1 2 3; do
  for tag in "$@"; do
    cp variants/lib_$tag.so dump1090_rs_amd/libadsb_hip.so
    for wl in sparse dense; do
      echo -n "$tag $wl: " >> $O
      timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --blocks 3 --workload $wl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms/step',d['ms_per_step'],'median',d['ms_per_step_median'],'blocks',d['ms_per_step_blocks']['all'],'kernel_alone',d['roofline']['kernel_avg_ms'],'device',d['roofline']['sustained']['device_ms_per_launch'],'parity',d.get('parity_checked'))" >> $O 2>&1
    done
  done
done
cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so
python tools/config1.py >> $O 2>&1
grep -v amdgpu $O
