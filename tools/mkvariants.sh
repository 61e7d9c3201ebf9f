#!/bin/bash
# Build A/B variants of libadsb_hip.so from the working tree: tools/mkvariants.sh tag1="-DFLAG=.." tag2="..."
# -> variants/lib_<tag>.so (git-ignored; the directory travels to the GPU box for tools/ab.sh: empty it when done).
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
for spec in "$@"; do
  tag=${spec%%=*}; flags=${spec#*=}; [ "$flags" = "$spec" ] && flags=""
  ADSB_HIPCC_FLAGS="$flags" python -m dump1090_rs_amd.build --force > /tmp/mkvar_$tag.log 2>&1 || { echo "build $tag failed"; tail -5 /tmp/mkvar_$tag.log; exit 1; }
  mkdir -p variants; cp dump1090_rs_amd/libadsb_hip.so variants/lib_$tag.so; echo "built $tag ($flags)"
done
