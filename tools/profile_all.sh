#!/bin/bash
# Everything profiles/ holds for one build, in one GPU call: kernel stats + traffic + bench line
# (tools/profile_round.sh), the same with blocking calls (tools/profile_sync.sh), and the bench lines
# of the other workloads.  usage: tools/profile_all.sh <tag>; then tools/save_profiles.py <tag> <name>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; [ -f "$R/bench.py" ] || { echo "no bench.py under $R" >&2; exit 1; }; T=$1; cd $R
./tools/profile_round.sh $T
./tools/profile_sync.sh $T
python3 tools/sq_counters.py sq_$T > gpurun_out/sq_$T.log 2>&1; tail -12 gpurun_out/sq_$T.log | head -8
for w in dense stream shard live; do
  timeout 600 python bench.py --workload $w --no-also > gpurun_out/bench_${w}_$T.json 2> gpurun_out/bench_${w}_$T.err
  tail -1 gpurun_out/bench_${w}_$T.json | cut -c1-200
done
python tools/hosttime.py ring --chunks 1 --depth 8 > gpurun_out/hosttime_ring_$T.txt 2>&1
for ch in 2 4 8 16; do python tools/hosttime.py ring --chunks $ch --depth 8 --passes 6000 >> gpurun_out/hosttime_ring_$T.txt 2>&1; done
python tools/hosttime.py ring --chunks 1 --depth 1 >> gpurun_out/hosttime_ring_$T.txt 2>&1
python tools/hosttime.py resident --chunks 1 --depth 8 >> gpurun_out/hosttime_ring_$T.txt 2>&1
python tools/experiments/config1.py > gpurun_out/config1_$T.txt 2>&1
ADSB_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --no-also > gpurun_out/bench_gloo2_$T.json 2> gpurun_out/bench_gloo2_$T.err
ADSB_BENCH_BACKEND=gloo timeout 1200 python bench.py --gpus 8 --no-also > gpurun_out/bench_gloo8_$T.json 2> gpurun_out/bench_gloo8_$T.err
ADSB_BENCH_BACKEND=gloo timeout 1200 python bench.py --gpus 8 --workload shard --capture-chunks 512 --no-also > gpurun_out/bench_gloo8_shard_$T.json 2> gpurun_out/bench_gloo8_shard_$T.err
ADSB_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --workload shard --no-also > gpurun_out/bench_gloo2_shard_$T.json 2> gpurun_out/bench_gloo2_shard_$T.err
tail -1 gpurun_out/bench_gloo2_$T.json | cut -c1-200; tail -1 gpurun_out/bench_gloo2_shard_$T.json | cut -c1-200
# the dense stream's kernels one by one (blocking calls: nothing overlaps)
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pdense_$T -o bench -- python3 $R/bench.py --sync --workload dense --steps 20 --warmup 4 --no-cpu-baseline --no-also > $R/gpurun_out/pdense_${T}_bench.log 2>&1
grep -h "adsb::" $R/gpurun_out/pdense_$T/bench_kernel_stats.csv | cut -d, -f1-4 | head -12
