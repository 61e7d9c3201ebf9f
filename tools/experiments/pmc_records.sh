#!/bin/bash
# VALU instructions of the dense stream's tail kernels per launch (blocking calls: a launch's counters are its own).
# usage: tools/pmc_records.sh <outdir-name>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; N=$1; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/$N -o p -- python3 $R/bench.py --sync --workload dense --steps 6 --warmup 4 --no-cpu-baseline --no-also --blocks 0 --ramp-ms 0 > /dev/null 2>&1
python3 - <<PY
import csv, collections, glob
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$R/gpurun_out/$N/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        import re
        m=re.search(r'(k_\w+(<[^>]*>)?)', r['Kernel_Name'])
        if m: agg[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
for name,c in sorted(agg.items()):
    print(name[:40].ljust(40), ' '.join(f"{k}={sum(v[-6:])/len(v[-6:]):.0f}" for k,v in sorted(c.items())), 'launches', len(next(iter(c.values()))))
PY
