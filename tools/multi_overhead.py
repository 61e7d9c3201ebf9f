"""The kernels of the LAST cluster of a rocprofv3 kernel trace (tools/multi_steps.py puts its timed steps between
two pauses): the union of their intervals = what the device was busy for; per kernel the count and mean duration.
usage: python tools/multi_overhead.py <kernel_trace.csv> <steps.json>"""
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = json.loads([l for l in open(sys.argv[2]).read().splitlines() if l.startswith("{")][-1])
who = "Thread_Id" if "Thread_Id" in rows[0] else "Queue_Id"     # the launching host thread = the device thread = the context
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r[who]) for r in rows), key=lambda x: x[0])
clusters, cur = [], [ks[0]]
for k in ks[1:]:
    if k[0] - max(k[1] for k in cur) > 30_000_000:   # a pause of more than 30 ms
        clusters.append(cur)
        cur = []
    cur.append(k)
clusters.append(cur)
import re
def short(name):
    m = re.search(r"\b(k_\w+)(<[^(]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name.split("(")[0][:60]
want_scans = steps["steps"] * steps["contexts"]
timed = next((c for c in reversed(clusters) if sum("k_scan_fast" in k[2] for k in c) == want_scans), clusters[-1])
busy, end = 0, 0
for s, e, *_ in timed:
    if e > end:
        busy += e - max(s, end)
        end = e
span = max(k[1] for k in timed) - timed[0][0]
per = {}
for s, e, name, _ in timed:
    per.setdefault(short(name), []).append(e - s)
n = steps["steps"]
out = {"steps": n, "contexts": steps["contexts"], "pipelined": steps["pipelined"],
       "ms_per_step_wall": steps["ms_per_step_wall"],
       "ms_per_step_device_busy": round(busy / n / 1e6, 4),
       "ms_per_step_first_kernel_to_last": round(span / n / 1e6, 4),
       "ms_overhead": round(steps["ms_per_step_wall"] - busy / n / 1e6, 4),
       "is": "device busy = union of the intervals of every kernel of the timed steps (rocprofv3 --kernel-trace); overhead = "
             "wall time per step minus that: launch latencies, the exchange between the phases, hand-overs between threads "
             "(the trace cannot say which context a kernel belongs to -- its Thread_Id is not the launching thread -- so the "
             "per-context 'device time' is the library's own host-clock figure: host_clock_stats.ms_phase*_max)",
       "host_clock_stats": steps["stats_mean"],
       "kernels_per_step": {k: {"launches": round(len(v) / n, 2), "mean_us": round(sum(v) / len(v) / 1e3, 2)} for k, v in sorted(per.items())}}
print(json.dumps(out, indent=1))
if len(sys.argv) > 3:   # the timed cluster, reduced: thread, kernel, start, end (ns from the cluster's first start)
    with open(sys.argv[3], "w") as f:
        for s_, e_, name, ctx in timed:
            f.write(f"{ctx},{short(name)},{s_ - timed[0][0]},{e_ - timed[0][0]}\n")
