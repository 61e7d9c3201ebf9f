#!/bin/bash
# Per-wave phase accounting of the scan (ADSB_KERNEL_ACCT build: variants/lib_acct.so; ADSB_DEBUG_STOP=100 ADSB_TIMELINE=2:
# every wave totals its clocks per phase and per barrier wait; printed for the LAST scan when the context is destroyed):
# the sparse and the dense workload, blocking and pipelined.
cd ${GRAFT_REPO_ROOT:-.}; T=${TAG:-acct}; mkdir -p gpurun_out
cp dump1090_rs_amd/libadsb_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so' EXIT   # whatever ends the script, the production library is back
cp variants/lib_acct.so dump1090_rs_amd/libadsb_hip.so
for w in sparse dense; do
  for mode in "--sync" ""; do
    echo "== $w ${mode:-pipelined}"
    ADSB_DEBUG_STOP=100 ADSB_TIMELINE=2 timeout 300 python bench.py --workload $w $mode --steps 12 --warmup 8 --blocks 0 --no-cpu-baseline --no-also 2>&1 >/dev/null | grep -A11 "phase accounting" | tail -12
  done
done
