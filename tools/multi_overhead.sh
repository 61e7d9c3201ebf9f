#!/bin/bash
# Orchestration overhead of adsb_multi_* from a kernel trace: tools/multi_overhead.sh <tag> [contexts] [chunks] [bursts per 512 buffers]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; T=$1; N=${2:-8}; CH=${3:-512}; B=${4:-64}
cd /tmp; export TMPDIR=/tmp
for mode in "" "--pipelined"; do
  tag=${T}_n${N}_c${CH}_b${B}${mode:+_pipe}
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/mo_$tag -o t -- python3 $R/tools/multi_steps.py --contexts $N --chunks $CH --steps 40 --bursts $B $mode > $R/gpurun_out/mo_$tag.json 2> $R/gpurun_out/mo_$tag.err
  f=$(find $R/gpurun_out/mo_$tag -name "t_kernel_trace.csv" | head -1)
  python3 $R/tools/multi_overhead.py $f $R/gpurun_out/mo_$tag.json $R/gpurun_out/mo_$tag.trace.csv > $R/gpurun_out/multi_overhead_$tag.json 2>> $R/gpurun_out/mo_$tag.err
  cat $R/gpurun_out/multi_overhead_$tag.json
  rm -rf $R/gpurun_out/mo_$tag
done
