#!/bin/bash
# One GPU session: the GPU suite, then the host-time breakdown of a one-buffer ring pass (tuning build,
# variants/lib_tune.so from tools/mkvariants.sh tune=-DADSB_TUNING) and the config-1 latencies.
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/${TAG:-s}_tests.log
cp dump1090_rs_amd/libadsb_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so' EXIT   # whatever ends the script, the production library is back
cp variants/lib_tune.so dump1090_rs_amd/libadsb_hip.so
for depth in 3 1; do
  ADSB_HOST_TIMES=1 python tools/hosttime.py ring --chunks 1 --depth $depth >> gpurun_out/${TAG:-s}_hosttime.log 2>&1
done
ADSB_HOST_TIMES=1 python tools/hosttime.py ring --chunks 4 --depth 3 --passes 8000 >> gpurun_out/${TAG:-s}_hosttime.log 2>&1
cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so
python tools/config1.py > gpurun_out/${TAG:-s}_config1.log 2>&1
tail -3 gpurun_out/${TAG:-s}_tests.log; cat gpurun_out/${TAG:-s}_hosttime.log gpurun_out/${TAG:-s}_config1.log
