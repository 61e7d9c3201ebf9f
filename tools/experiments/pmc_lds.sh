#!/bin/bash
# Which phase of k_scan_fast the LDS bank conflicts belong to: SQ LDS counters with the kernel cut short after
# P1 / P2 / P3 / P4 / P5 / whole (ADSB_DEBUG_STOP: needs a -DADSB_TUNING build in place).  Blocking launches
# (--sync) so that a launch's counters are its own.   usage: tools/pmc_lds.sh <outdir-name>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; [ -f "$R/bench.py" ] || { echo "no bench.py under $R" >&2; exit 1; }; N=$1; cd /tmp; export TMPDIR=/tmp
for stop in 1 2 3 4 5 0; do
  export ADSB_DEBUG_STOP=$stop
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/$N/s$stop -o p -- python3 $R/bench.py --sync --steps 3 --warmup 1 --no-cpu-baseline --no-also --buffers 2 --blocks 0 --ramp-ms 0 > /dev/null 2>&1
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
print('cut after %s:' % ({'1':'P1','2':'P2','3':'P3','4':'P4','5':'P5','0':'whole'}['$stop']), ' '.join(f"{k}={sum(v)/len(v):.0f}" for k,v in sorted(agg.items())), 'launches=%d us=%.1f'%(len(dur), sum(dur)/max(1,len(dur))))
PY
done
