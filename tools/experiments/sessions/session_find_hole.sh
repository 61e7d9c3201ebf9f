cd ${GRAFT_REPO_ROOT:-.}
cp dump1090_rs_amd/libadsb_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so' EXIT
cp variants/lib_oldfresh.so dump1090_rs_amd/libadsb_hip.so
for seed in 27182 27183 27184 27185; do timeout 400 python tests/fuzz_gpu.py --cases 1 --dense 0 --mixed 0 --multi 120 --seed $seed 2>&1 | grep "MISMATCH\|identical" | cut -c1-200; done
