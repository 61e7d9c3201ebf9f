#!/bin/bash
# GPU suite + the driver's bench command; results under gpurun_out/<TAG>_*.
T=${TAG:-s}; mkdir -p gpurun_out
if [ "${TESTS:-1}" = "1" ]; then
  timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/${T}_tests.log; tail -4 gpurun_out/${T}_tests.log
fi
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 ${BENCH_ARGS} > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
tail -3 gpurun_out/${T}_bench.err; python - <<PY
import json
d=json.load(open("gpurun_out/${T}_bench.json"))
print("value",d["value"],"ms/step",d["ms_per_step"],"blocks",d.get("ms_per_step_blocks"),"frac",d["roofline"]["frac"],"parity",d.get("parity_checked"))
print("cpu",d.get("cpu_baseline",{}).get("value"),(d.get("cpu_baseline",{}).get("all_cores") or {}).get("value"))
a=d.get("also",{})
c1=a.get("config1_cargo_bench_case",{}); print("config1",{k:v for k,v in c1.items() if k.startswith("ms_") or k=="parity_checked"})
c3=a.get("config3_streaming_ring",{}); print("config3",[(x["buffers_per_slot"],x["value"],x["parity_checked"]) for x in c3.get("slot_sweep",[])])
c5=a.get("config5_dense",{}); print("config5",c5.get("ms_per_step"),c5.get("ms_per_step_median"),c5.get("parity_checked"))
PY
