#!/bin/bash
# Round 6 session F: the HOST side of the whole library (the eight .cpp units, compiled by g++ with -fsanitize=address,undefined;
# the three .hip units as always) under the GPU suite on the GPU box.  Not a device sanitizer: the code objects are the ordinary
# ones, only host code is instrumented.  variants/lib_hostasan.so is built by tools/mk_hostasan.sh.
# (libstdc++ is preloaded beside libasan: python loads it late, and ASan's __cxa_throw interceptor must find the real one at start-up.)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6_f
mkdir -p $O
cp dump1090_rs_amd/libadsb_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so' EXIT
cp variants/lib_hostasan.so dump1090_rs_amd/libadsb_hip.so
PRE="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so.6)"
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=0
export UBSAN_OPTIONS=print_stacktrace=1
# (torch dlopen()s its own libraries by RPATH, which the sanitizer's dlopen interceptor does not carry over)
export LD_LIBRARY_PATH=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))"):$LD_LIBRARY_PATH
LD_PRELOAD="$PRE" timeout 300 python -c "import torch; x = torch.zeros(10, device='cuda'); print('torch under the preload', x.sum().item())" 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-300
LD_PRELOAD="$PRE" timeout 2700 python -m pytest tests/test_gpu_multi.py tests/test_gpu_small_pass.py tests/test_gpu_shard8.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider \
   -k "not bench and not rccl and not compiled_c_host and not feed_tool" > $O/pytest_hostasan.log 2>&1; echo "pytest rc=$?"
grep -v "^    #" $O/pytest_hostasan.log | tail -12 | cut -c1-300
echo "AddressSanitizer reports: $(grep -c 'ERROR: AddressSanitizer' $O/pytest_hostasan.log)   UBSan reports: $(grep -c 'runtime error' $O/pytest_hostasan.log)"
grep -m3 -A14 "ERROR: AddressSanitizer\|runtime error" $O/pytest_hostasan.log | cut -c1-250 | head -60
