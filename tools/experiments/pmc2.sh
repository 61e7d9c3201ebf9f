#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; [ -f "$R/bench.py" ] || { echo "no bench.py under $R" >&2; exit 1; }; N=$1; shift; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_LDS_[A-Z_]*\|SQ_INST_CYCLES_[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST_[A-Z_]*\|SQ_INSTS_[A-Z_]*" | sort -u | tr '\n' ' ' | head -c 1500; echo
i=0
for set in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/$N/s$i -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also --buffers 2 > /dev/null 2>&1
  python3 - <<PY
import csv, collections
rows=list(csv.DictReader(open('$R/gpurun_out/$N/s$i/p_counter_collection.csv')))
agg=collections.defaultdict(list)
for r in rows:
    if 'k_scan_fast' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()): print(f"{k:28s} {sum(v)/len(v):14.0f}")
PY
done
