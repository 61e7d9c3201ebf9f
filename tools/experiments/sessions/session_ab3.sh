#!/bin/bash
T=${TAG:-s}; mkdir -p gpurun_out; O=gpurun_out/${T}_ab.log
cp dump1090_rs_amd/libadsb_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so' EXIT   # whatever ends the script, the production library is back
fmt='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("ms/step",d["ms_per_step"],"median",d["ms_per_step_median"],"kernel_alone",d["roofline"]["kernel_avg_ms"],"device",d["roofline"]["sustained"]["device_ms_per_launch"])'
for rep in 1 2 3; do
  for wl in sparse dense; do
    echo -n "r3 $wl: " >> $O; ( cd variants/r3tree && timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --workload $wl 2>/dev/null | python -c "$fmt" ) >> $O 2>&1
    for tag in "$@"; do
      cp variants/lib_$tag.so dump1090_rs_amd/libadsb_hip.so
      echo -n "$tag $wl: " >> $O
      timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --blocks 0 --workload $wl 2>/dev/null | python -c "$fmt" >> $O 2>&1
    done
  done
done
for tag in "$@"; do
  cp variants/lib_$tag.so dump1090_rs_amd/libadsb_hip.so
  echo "== tail kernels, $tag" >> $O; ./tools/pmc_records.sh pmcrec_${T}_$tag >> $O 2>&1
done
cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so
python tools/config1.py >> $O 2>&1
grep -v amdgpu $O
