#!/bin/bash
# Round 6 session G: the host side under ThreadSanitizer over adsb_multi's GPU tests (real shard backend, real device threads).
# (setarch -R: this libtsan does not know the box's randomised address-space layout; env sets the preload after that, before python --
# and with it anything that touches the GPU -- starts)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6_g
mkdir -p $O
cp dump1090_rs_amd/libadsb_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so' EXIT
cp variants/lib_hosttsan.so dump1090_rs_amd/libadsb_hip.so
PRE="$(gcc -print-file-name=libtsan.so) $(gcc -print-file-name=libstdc++.so.6)"
export TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0:history_size=4
export LD_LIBRARY_PATH=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))"):$LD_LIBRARY_PATH
timeout 300 setarch x86_64 -R env LD_PRELOAD="$PRE" python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; grep -v "^    #" $O/smoke.log | tail -4 | cut -c1-300
timeout 1500 setarch x86_64 -R env LD_PRELOAD="$PRE" python -m pytest tests/test_gpu_multi.py -q -m gpu -p no:cacheprovider -k "not compiled_c_host and not two_gib" > $O/pytest_hosttsan.log 2>&1; echo "pytest rc=$?"
grep -v "^    #" $O/pytest_hosttsan.log | tail -8 | cut -c1-300
echo "ThreadSanitizer reports: $(grep -c 'WARNING: ThreadSanitizer' $O/pytest_hosttsan.log)"
grep "SUMMARY: ThreadSanitizer" $O/pytest_hosttsan.log | sort | uniq -c | sort -rn | head -15 | cut -c1-250
