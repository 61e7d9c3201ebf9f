"""ADSB_DEBUG_STOP=n ablation of k_scan_fast (needs a library built with -DADSB_TUNING): the kernel cut
short after P1 / P2 / P3 patterns / 6: + compaction / P4 gates / P5 (no epilogue) / whole.  Extra arguments go to bench.py
(e.g. --sync).  usage: python tools/ablate.py [--sync]"""
import json
import os
import subprocess
import sys

for stop in (1, 2, 3, 6, 4, 5, 0):
    env = dict(os.environ, ADSB_DEBUG_STOP=str(stop))
    out = subprocess.run([sys.executable, 'bench.py', '--steps', '20', '--warmup', '3', '--no-cpu-baseline', '--no-also',
                          '--buffers', '2', *sys.argv[1:]], env=env, capture_output=True, text=True).stdout.strip().splitlines()
    r = json.loads(out[-1])
    print(stop, r['roofline']['kernel_avg_ms'], r['ms_per_step'], flush=True)
