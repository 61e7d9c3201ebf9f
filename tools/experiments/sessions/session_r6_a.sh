#!/bin/bash
# Round 6 session A: the new adsb_multi GPU tests, then the scan streams of small contexts with a hardware queue each
# (ADSB_POOL_SMALL=5, hipExtStreamCreateWithCUMask) against four normal-priority pool streams (=2) in the bench's own
# order of legs, with the runtime's queue log; then the production library through the whole bench line.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6_a
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_multi.py -x -q -m gpu > $O/pytest_multi.log 2>&1; echo "pytest multi rc=$?"; tail -5 $O/pytest_multi.log
timeout 600 python -m pytest tests/test_gpu_multidevice.py tests/test_gpu_parity.py -x -q -m gpu -k "rccl or driver_contract or behind_the_ranks" > $O/pytest_bench.log 2>&1; echo "pytest bench rc=$?"; tail -5 $O/pytest_bench.log
cp dump1090_rs_amd/libadsb_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so' EXIT
cp variants/lib_tune.so dump1090_rs_amd/libadsb_hip.so
for rep in 1 2; do
for v in 2 5; do
  for m in "" early0; do
    echo -n "ADSB_POOL_SMALL=$v ${m:-bench_order}: "
    ADSB_POOL_SMALL=$v AMD_LOG_LEVEL=3 AMD_LOG_MASK=16 timeout 400 python tools/ring_history_probe.py $m 2> $O/q_${v}_${m:-bench}_$rep.log | tail -1
    grep -E "Created SWq|Selected queue" $O/q_${v}_${m:-bench}_$rep.log | sed 's/.*us: *//' > $O/q_${v}_${m:-bench}_$rep.queues.txt; rm $O/q_${v}_${m:-bench}_$rep.log
  done
done
done
cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -3 $O/bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6_a/bench_default.json"))
a = d["also"]
print("sparse", d["ms_per_step"], "dense", a["config5_dense"]["ms_per_step"], a["config5_dense"]["ms_per_step_blocks"])
print("ring", [(x["buffers_per_slot"], x["value"]) for x in a["config3_streaming_ring"]["slot_sweep"]])
print("config1", {k: v for k, v in a["config1_cargo_bench_case"].items() if k.startswith("ms_")})
for r in a["config4_sharded_capture"]["runs"]:
    print("config4", r["sky"], r["shards"], r["value"], r["ms_per_step"], r["roofline"]["frac"], r["parity_checked"], r["blocking_steps_host_clock"]["ms_wall"], r["wait"])
PY
