#!/bin/bash
# One-stop profile of the current build on the GPU box: rocprofv3 kernel stats, HBM traffic
# counters, final bench line.  usage: tools_profile_round.sh <tag>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; [ -f "$R/bench.py" ] || { echo "no bench.py under $R" >&2; exit 1; }; T=$1; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$T -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also > $R/gpurun_out/prof_${T}_bench.log 2>&1
cd $R; ./tools/traffic.sh traffic_$T > gpurun_out/traffic_$T.json 2>/dev/null
timeout 600 python bench.py > gpurun_out/bench_$T.json 2> gpurun_out/bench_$T.err
tail -1 gpurun_out/traffic_$T.json; tail -1 gpurun_out/bench_$T.json | cut -c1-400
