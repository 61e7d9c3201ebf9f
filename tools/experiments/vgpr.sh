#!/bin/bash
# VGPR / spill report of k_scan_fast at a given waves-per-SIMD target: tools/vgpr.sh [4|5]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
W=${1:-4}; D=/tmp/isa; mkdir -p $D; cp $R/dump1090_rs_amd/csrc/adsb_*.h $D/
sed "s/constexpr int kWavesPerSimd = kThreads == 512 ? 8 : [0-9];/constexpr int kWavesPerSimd = kThreads == 512 ? 8 : $W;/" $R/dump1090_rs_amd/csrc/adsb_scan_fast.hip > $D/x.hip
cd $D && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -x hip -S --cuda-device-only -o x$W.s x.hip 2>&1 | grep -c "failed to meet"
grep "\.vgpr_count\|vgpr_spill\|sgpr_spill\|group_segment_fixed_size:" x$W.s
