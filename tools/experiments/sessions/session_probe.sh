#!/bin/bash
T=${TAG:-s}; mkdir -p gpurun_out
python tools/clockprobe.py > gpurun_out/${T}_clockprobe.log 2>&1
grep -v amdgpu gpurun_out/${T}_clockprobe.log
( cd variants/r3tree && timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --workload dense 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r3 dense ms/step',d['ms_per_step'],'median',d['ms_per_step_median'],'device',d['roofline']['sustained']['device_ms_per_launch'])" )
timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --blocks 0 --workload dense 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('now dense ms/step',d['ms_per_step'],'median',d['ms_per_step_median'],'device',d['roofline']['sustained']['device_ms_per_launch'])"
( cd variants/r3tree && timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --workload dense 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r3 dense ms/step',d['ms_per_step'],'median',d['ms_per_step_median'],'device',d['roofline']['sustained']['device_ms_per_launch'])" )
timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also --blocks 0 --workload dense 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('now dense ms/step',d['ms_per_step'],'median',d['ms_per_step_median'],'device',d['roofline']['sustained']['device_ms_per_launch'])"
