#!/bin/bash
# One GPU session for the one-launch small pass: its tests (FULL=1: the whole GPU suite), the host-time
# breakdown of a one-buffer ring pass (tuning build: variants/lib_tune.so) and the config-1 latencies.
T=${TAG:-s}; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_small_pass.py -x -q 2>&1 | tail -25 > gpurun_out/${T}_small_tests.log
tail -4 gpurun_out/${T}_small_tests.log
if [ "${FULL:-0}" = "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/${T}_tests.log; tail -4 gpurun_out/${T}_tests.log
fi
cp dump1090_rs_amd/libadsb_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so' EXIT   # whatever ends the script, the production library is back
python tools/config1.py > gpurun_out/${T}_config1.log 2>&1
H=gpurun_out/${T}_hosttime_rel.log
for prof in 1 0; do for depth in 4 6 8; do
  python tools/hosttime.py ring --chunks 1 --depth $depth --profiling $prof >> $H 2>&1
done; done
python tools/hosttime.py ring --chunks 1 --depth 1 --profiling 0 >> $H 2>&1
python tools/hosttime.py ring --chunks 2 --depth 8 --profiling 0 --passes 10000 >> $H 2>&1
python tools/hosttime.py ring --chunks 4 --depth 8 --profiling 0 --passes 8000 >> $H 2>&1
python tools/hosttime.py ring --chunks 8 --depth 8 --profiling 0 --passes 5000 >> $H 2>&1
python tools/hosttime.py ring --chunks 16 --depth 8 --profiling 0 --passes 3000 >> $H 2>&1
python tools/hosttime.py resident --chunks 1 --depth 8 --profiling 0 >> $H 2>&1
python tools/hosttime.py resident --chunks 1 --depth 1 --profiling 0 >> $H 2>&1
cp variants/lib_tune.so dump1090_rs_amd/libadsb_hip.so
H=gpurun_out/${T}_hosttime.log
ADSB_HOST_TIMES=1 python tools/hosttime.py ring --chunks 1 --depth 8 --profiling 0 >> $H 2>&1
ADSB_HOST_TIMES=1 python tools/hosttime.py resident --chunks 1 --depth 8 --profiling 0 >> $H 2>&1
for ch in 8 16; do
  ADSB_RING_COPY=1 python tools/hosttime.py ring --chunks $ch --depth 8 --profiling 0 --passes 4000 2>&1 | sed "s/^/copy, then one launch: /" >> $H
done
cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so
grep -v amdgpu.ids gpurun_out/${T}_hosttime_rel.log gpurun_out/${T}_hosttime.log gpurun_out/${T}_config1.log
