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
# the config-4 leg of the default line under a kernel trace (round 6): profiles/<name>_config4_kernel_stats.csv
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pconfig4_$T -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --also-only config4 > $R/gpurun_out/pconfig4_${T}_bench.log 2>&1
cd $R; grep "adsb::" gpurun_out/pconfig4_$T/bench_kernel_stats.csv | cut -d, -f1-4 | head -12
