"""Kernel-trace of the one-buffer ring: how long each one-launch pass's kernel runs and how many run side by side.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ringtrace -- python3 tools/hosttime.py ring --chunks 1 --depth 8 --passes 4000
    python tools/ring_trace_stats.py gpurun_out/ringtrace
"""
import csv, glob, sys
import numpy as np
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_scan_fast" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r.get("Queue_Id", 0) or 0)))
rows.sort()
rows = rows[len(rows) // 4:]          # steady state
s = np.array([r[0] for r in rows], dtype=np.float64)
e = np.array([r[1] for r in rows], dtype=np.float64)
d = (e - s) / 1e3
print(f"{len(rows)} launches: duration mean {d.mean():.1f} us, median {np.median(d):.1f}, p10 {np.percentile(d,10):.1f}, p90 {np.percentile(d,90):.1f}")
print(f"start-to-start {np.diff(s).mean()/1e3:.2f} us; queues used: {len(set(r[2] for r in rows))}")
# concurrency: time-weighted number of kernels running
ev = sorted([(t, 1) for t in s] + [(t, -1) for t in e])
run, last, acc = 0, ev[0][0], np.zeros(16)
for t, k in ev:
    acc[run] += t - last
    last = t
    run += k
tot = acc.sum()
print("share of time with n kernels running:", " ".join(f"{n}:{acc[n]/tot:.2f}" for n in range(9)))
