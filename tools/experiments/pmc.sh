#!/bin/bash
# usage: tools_pmc.sh <outdir-name>   (run on the GPU box from the repo root)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; [ -f "$R/bench.py" ] || { echo "no bench.py under $R" >&2; exit 1; }; N=$1; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/gpurun_out/$N/a -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also --buffers 2 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/$N/b -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also --buffers 2 > /dev/null 2>&1
python3 - <<PY
import csv, collections
for d in ('a','b'):
    rows=list(csv.DictReader(open('$R/gpurun_out/$N/'+d+'/p_counter_collection.csv')))
    agg=collections.defaultdict(list)
    for r in rows:
        if 'k_scan_fast' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print(f"{k:24s} {sum(v)/len(v):14.0f}")
PY
