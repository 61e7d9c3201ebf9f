#!/bin/bash
# Which phase the scan's LDS bank conflicts belong to (tuning build: variants/lib_tune.so).
T=${TAG:-s}; mkdir -p gpurun_out
cp dump1090_rs_amd/libadsb_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so' EXIT   # whatever ends the script, the production library is back
cp variants/lib_tune.so dump1090_rs_amd/libadsb_hip.so
./tools/pmc_lds.sh pmclds_$T > gpurun_out/${T}_pmc_lds.txt 2>&1
cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so
cat gpurun_out/${T}_pmc_lds.txt
