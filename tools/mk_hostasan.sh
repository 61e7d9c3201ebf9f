#!/bin/bash
# variants/lib_hostasan.so: libadsb_hip.so with its HOST units (csrc/*.cpp) compiled by g++ under -fsanitize=address,undefined and
# its kernels (csrc/*.hip) compiled by hipcc as always -- for running the GPU suite with the host code instrumented
# (tools/experiments/sessions/session_r6_f.sh: LD_PRELOAD=libasan.so python -m pytest ...).  Not a device sanitizer.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/dump1090_rs_amd/csrc; W=$(mktemp -d)
for f in adsb_scan_fast adsb_scan_simple adsb_aux; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c $C/$f.hip -o $W/$f.o; done
for f in adsb_context adsb_pass adsb_collect adsb_ring adsb_shard adsb_multi adsb_selftest adsb_replay_host; do
  g++ -std=c++17 -O1 -g -fPIC -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c $C/$f.cpp -o $W/$f.o
done
mkdir -p $R/variants
g++ -shared -fsanitize=address,undefined -o $R/variants/lib_hostasan.so $W/*.o -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
rm -rf $W; ls -la $R/variants/lib_hostasan.so
