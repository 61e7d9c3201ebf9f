#!/bin/bash
# One GPU session for the one-launch small pass: its tests, the whole GPU suite, then the host-time
# breakdown of a one-buffer ring pass (tuning build: variants/lib_tune.so) and the config-1 latencies.
T=${TAG:-s}; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_small_pass.py -x -q 2>&1 | tail -25 > gpurun_out/${T}_small_tests.log
tail -4 gpurun_out/${T}_small_tests.log
if [ "${FULL:-1}" = "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/${T}_tests.log; tail -4 gpurun_out/${T}_tests.log
fi
cp dump1090_rs_amd/libadsb_hip.so /tmp/rel.so
python tools/config1.py > gpurun_out/${T}_config1.log 2>&1
for prof in 1 0; do for depth in 3 4 1; do
  python tools/hosttime.py ring --chunks 1 --depth $depth --profiling $prof >> gpurun_out/${T}_hosttime_rel.log 2>&1
done; done
python tools/hosttime.py ring --chunks 4 --depth 3 --passes 8000 >> gpurun_out/${T}_hosttime_rel.log 2>&1
python tools/hosttime.py ring --chunks 16 --depth 3 --passes 4000 >> gpurun_out/${T}_hosttime_rel.log 2>&1
cp variants/lib_tune.so dump1090_rs_amd/libadsb_hip.so
for depth in 3 1; do
  ADSB_HOST_TIMES=1 python tools/hosttime.py ring --chunks 1 --depth $depth >> gpurun_out/${T}_hosttime.log 2>&1
done
ADSB_RING_COPY=1 ADSB_HOST_TIMES=1 python tools/hosttime.py ring --chunks 1 --depth 3 >> gpurun_out/${T}_hosttime.log 2>&1
ADSB_NO_FUSE=1 ADSB_HOST_TIMES=1 python tools/hosttime.py ring --chunks 1 --depth 3 >> gpurun_out/${T}_hosttime.log 2>&1
cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so
grep -v amdgpu.ids gpurun_out/${T}_hosttime_rel.log gpurun_out/${T}_hosttime.log gpurun_out/${T}_config1.log
