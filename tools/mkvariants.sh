#!/bin/bash
# Build A/B variants of libadsb_hip.so from the working tree: tools/mkvariants.sh tag1="-DFLAG=.." tag2="..."
# -> gpurun_lib_<tag>.so in the repo root (git-ignored; they travel to the GPU box for tools/ab.sh).
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
for spec in "$@"; do
  tag=${spec%%=*}; flags=${spec#*=}; [ "$flags" = "$spec" ] && flags=""
  ADSB_HIPCC_FLAGS="$flags" python -m dump1090_rs_amd.build --force > /tmp/mkvar_$tag.log 2>&1 || { echo "build $tag failed"; tail -5 /tmp/mkvar_$tag.log; exit 1; }
  cp dump1090_rs_amd/libadsb_hip.so gpurun_lib_$tag.so; echo "built $tag ($flags)"
done
