#!/bin/bash
# Round 6, VERDICT r5 item 4: which hardware queue does every stream of the process land on, in the bench's order and with
# a small context created first?  ROCclr logs queue creation / selection under AMD_LOG_MASK bit 4 (LOG_QUEUE).
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6_queues
mkdir -p $O
for m in "" early0; do
  tag=${m:-bench_order}
  AMD_LOG_LEVEL=3 AMD_LOG_MASK=16 timeout 400 python tools/ring_history_probe.py $m > $O/$tag.out 2> $O/$tag.log
  tail -1 $O/$tag.out
  grep -c . $O/$tag.log
  grep -iE "queue|HWq|SWq" $O/$tag.log | head -60 > $O/$tag.queues.txt
  # keep the merged-back log small
  head -c 400000 $O/$tag.log > $O/$tag.log.head; rm $O/$tag.log
done
# the dense leg alone in a fresh process, and the driver's line (with the small_context_first block as it is)
AMD_LOG_LEVEL=3 AMD_LOG_MASK=16 timeout 300 python bench.py --workload dense --steps 200 --no-also --no-cpu-baseline > $O/dense_alone.json 2> $O/dense_alone.log
grep -iE "queue|HWq|SWq" $O/dense_alone.log | head -40 > $O/dense_alone.queues.txt; rm $O/dense_alone.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6_queues/dense_alone.json"))
print("dense alone", d["ms_per_step"], d.get("ms_per_step_blocks"))
PY
AMD_LOG_LEVEL=3 AMD_LOG_MASK=16 timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.log
grep -iE "queue|HWq|SWq" $O/bench_default.log | head -80 > $O/bench_default.queues.txt; rm $O/bench_default.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r6_queues/bench_default.json"))
print("default: sparse", d["ms_per_step"], "dense", d["also"]["config5_dense"]["ms_per_step"], d["also"]["config5_dense"]["ms_per_step_blocks"],
      "ring", [(x["buffers_per_slot"], x["value"]) for x in d["also"]["config3_streaming_ring"]["slot_sweep"]])
PY
