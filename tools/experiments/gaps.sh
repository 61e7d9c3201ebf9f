#!/bin/bash
# gaps between consecutive k_scan_fast launches (end -> next start) from a rocprofv3 kernel trace
# usage: tools/gaps.sh <tag> [bench args...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; [ -f "$R/bench.py" ] || { echo "no bench.py under $R" >&2; exit 1; }; T=$1; shift; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gaps_$T -o g -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also "$@" > $R/gpurun_out/gaps_$T.log 2>&1
python3 - <<PY
import csv, glob
f=glob.glob('$R/gpurun_out/gaps_$T/**/g_kernel_trace.csv', recursive=True)[0]
k=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp'])) for r in csv.DictReader(open(f)) if 'k_scan_fast' in r['Kernel_Name'])
gaps=sorted((k[i+1][0]-k[i][1])/1e3 for i in range(len(k)-1)); durs=sorted((e-s)/1e3 for s,e in k)
print('$T', 'n', len(k), 'median gap us', gaps[len(gaps)//2], 'median dur us', durs[len(durs)//2])
PY
tail -1 $R/gpurun_out/gaps_$T.log | cut -c1-130
