#!/bin/bash
# Everything profiles/r5_v20_* holds, in one GPU call: tools/profile_all.sh, then what round 5 added -- adsb_multi_* from one
# process (N = 1 and eight contexts on the one GPU), the small-pass host times with a flush per pass, the stage split of the scan
# (needs variants/lib_tune.so of the same source).   usage: tools/profile_r5.sh <tag>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; T=$1; cd $R
./tools/profile_all.sh $T
for n in 1 8; do
  timeout 600 python bench.py --workload shard --single-process --contexts $n > gpurun_out/bench_multi${n}_$T.json 2> gpurun_out/bench_multi${n}_$T.err
  tail -1 gpurun_out/bench_multi${n}_$T.json | cut -c1-240
done
python tools/hosttime.py resident --chunks 1 --depth 8 2>&1 | grep -v amdgpu.ids > gpurun_out/hosttime_resident_$T.txt; cat gpurun_out/hosttime_resident_$T.txt
tools/experiments/sessions/session_ablate.sh 2>&1 | grep -v amdgpu.ids > gpurun_out/ablate_$T.txt; rm -rf gpurun_out/abl_r5; tail -9 gpurun_out/ablate_$T.txt
