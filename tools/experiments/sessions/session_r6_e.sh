#!/bin/bash
# Round 6 session E: final tree -- the GPU suite, the driver's command, a kernel trace of the config-4 leg, a second soak seed.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6_e
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -3 $O/pytest_gpu.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err ) 2>&1 | grep real; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6_e/bench_driver_cmd.json"))
a = d["also"]
print("sparse", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], "dense", a["config5_dense"]["ms_per_step"],
      "ring", a["config3_streaming_ring"]["slot_sweep"][0]["value"], "errors", [k for k, v in a.items() if "error" in v])
for r in a["config4_sharded_capture"]["runs"]:
    print("config4", r["sky"], r["shards"], r["value"], r["ms_per_step"], r["roofline"]["frac"], r["parity_checked"], r["wait"])
PY
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_config4 -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --also-only config4 > $R/$O/prof_config4_bench.log 2>&1
cd $R
grep "adsb::" $O/prof_config4/bench_kernel_stats.csv | cut -d, -f1-7 | head -20
( time timeout 1500 python tests/fuzz_gpu.py --cases 1200 --seed 20261003 --max-chunks 24 --dense 40 --mixed 120 --multi 500 ) > $O/soak2.txt 2>&1; echo "soak rc=$?"; grep -v amdgpu.ids $O/soak2.txt | tail -6 | cut -c1-1200
