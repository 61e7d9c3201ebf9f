#!/bin/bash
# Per-phase instruction counts of k_scan_fast: SQ counters with the kernel cut short after
# P1 / P2 / P3 patterns / + compaction (6) / P4 gates / (all but the epilogue) / whole.  Needs a -DADSB_TUNING library installed.  usage: tools/pmc_ablate.sh <outdir-name>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; [ -f "$R/bench.py" ] || { echo "no bench.py under $R" >&2; exit 1; }; N=$1; cd /tmp; export TMPDIR=/tmp
for stop in 1 2 3 6 4 5 0; do
  export ADSB_DEBUG_STOP=$stop
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/$N/s$stop -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also --buffers 2 > /dev/null 2>&1
  python3 - <<PY
import csv, collections, glob
agg=collections.defaultdict(list)
for f in glob.glob('$R/gpurun_out/$N/s$stop/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_scan_fast' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
dur=[]
for f in glob.glob('$R/gpurun_out/$N/s$stop/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_scan_fast' in r['Kernel_Name']: dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
print('stop=$stop', ' '.join(f"{k}={sum(v)/len(v):.0f}" for k,v in sorted(agg.items())), 'us=%.1f'%(sum(dur)/max(1,len(dur))))
PY
done
