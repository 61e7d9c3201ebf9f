#!/bin/bash
# Timeline of the pipelined bench from a rocprofv3 kernel trace: per step the start / end of
# scan, match and records relative to the previous scan's start.  usage: [TL_FROM=3 TL_TO=9] tools/timeline.sh <tag> [bench args]  (window = scans FROM..TO)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; [ -f "$R/bench.py" ] || { echo "no bench.py under $R" >&2; exit 1; }; T=$1; shift; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$T -o g -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-also "$@" > $R/gpurun_out/tl_$T.log 2>&1
python3 - <<PY
import csv, glob
f=glob.glob('$R/gpurun_out/tl_$T/**/g_kernel_trace.csv', recursive=True)[0]
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name']) for r in csv.DictReader(open(f)) if 'adsb::' in r['Kernel_Name'] or 'fillBuffer' in r['Kernel_Name']]
rows.sort()
def short(n):
    for k in ('k_scan_fast','k_match','k_records','k_order_prefix','k_order_scatter','k_order_rank','k_score','k_emit','fillBuffer'):
        if k in n: return k
    return n[:20]
scans=[r for r in rows if 'k_scan_fast' in r[2]]
A,B=int('${TL_FROM:-3}'),int('${TL_TO:-9}')
t0=scans[A][0]
for s,e,n in rows:
    if s < scans[A][0] or s > scans[B][1]: continue
    print(f"{short(n):12s} start {(s-t0)/1e3:9.1f}  end {(e-t0)/1e3:9.1f}  dur {(e-s)/1e3:7.1f}")
iv=[(scans[i+1][0]-scans[i][0])/1e3 for i in range(3,len(scans)-1)]
print('scan start-to-start intervals us:', [round(x,1) for x in iv])
PY
tail -1 $R/gpurun_out/tl_$T.log | cut -c1-160
