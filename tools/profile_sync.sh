#!/bin/bash
# rocprofv3 kernel stats + bench line with blocking calls (launches do not overlap: a launch's
# duration is the kernel alone).  usage: tools/profile_sync.sh <tag>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; [ -f "$R/bench.py" ] || { echo "no bench.py under $R" >&2; exit 1; }; T=$1; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/psync_$T -o bench -- python3 $R/bench.py --sync --steps 20 --warmup 3 --no-cpu-baseline --no-also > $R/gpurun_out/psync_${T}_bench.log 2>&1
cd $R; timeout 300 python bench.py --sync --no-cpu-baseline --no-also > gpurun_out/bench_sync_$T.json 2> /dev/null
grep "k_scan_fast" gpurun_out/psync_$T/bench_kernel_stats.csv | cut -c1-140; tail -1 gpurun_out/bench_sync_$T.json | cut -c1-300
