#!/bin/bash
# Round 5: the scan cut short after each stage (tuning build): SQ counters per cut (tools/experiments/pmc_ablate.sh) and clean blocking durations (tools/ablate.py --sync)
cd ${GRAFT_REPO_ROOT:-.}
cp dump1090_rs_amd/libadsb_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so' EXIT   # whatever ends the script, the production library is back
cp variants/lib_tune.so dump1090_rs_amd/libadsb_hip.so
echo "== SQ counters per launch of k_scan_fast, kernel cut after stage (ADSB_DEBUG_STOP)"; tools/experiments/pmc_ablate.sh abl_r5
echo "== blocking launches, no counters: stop, kernel_avg_ms, ms_per_step"; python tools/ablate.py --sync
