#!/bin/bash
# HBM traffic of k_scan_fast from PMC counters (separate passes), per launch.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; [ -f "$R/bench.py" ] || { echo "no bench.py under $R" >&2; exit 1; }; N=$1; cd /tmp; export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/$N/$ctr -o p -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-also > /dev/null 2>&1
done
python3 - <<PY
import csv, collections, json
out={}
for ctr in ('FETCH_SIZE','WRITE_SIZE'):
    rows=list(csv.DictReader(open('$R/gpurun_out/$N/'+ctr+'/p_counter_collection.csv')))
    vals=collections.defaultdict(list)
    for r in rows:
        if r['Counter_Name']==ctr:
            k='scan' if 'k_scan_fast' in r['Kernel_Name'] else ('match' if 'k_match' in r['Kernel_Name'] else None)
            if k: vals[k].append(float(r['Counter_Value']))
    for k,v in vals.items(): out[f'{k}_{ctr}_avg']=sum(v)/len(v); out[f'{k}_{ctr}_n']=len(v)
print(json.dumps(out))
PY
