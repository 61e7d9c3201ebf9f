#!/bin/bash
# Round 6 session B: the whole GPU suite on library 0.21; adsb_multi's wait policy under taskset; the large contexts' streams
# with a hardware queue each (tuning build, ADSB_POOL_LARGE) against the priority pools, sparse and dense, twice.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6_b
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -4 $O/pytest_gpu.log
./tools/experiments/sessions/session_r6_wait.sh
cp gpurun_out/r6_wait.txt $O/
cp dump1090_rs_amd/libadsb_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so' EXIT
cp variants/lib_tune.so dump1090_rs_amd/libadsb_hip.so
for rep in 1 2; do
for v in 0 1 2; do
  echo -n "ADSB_POOL_LARGE=$v sparse: "; ADSB_POOL_LARGE=$v timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_blocks']['all'], d['roofline']['kernel_avg_ms'])"
  echo -n "ADSB_POOL_LARGE=$v dense:  "; ADSB_POOL_LARGE=$v timeout 300 python bench.py --workload dense --steps 200 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_blocks']['all'])"
done
done
