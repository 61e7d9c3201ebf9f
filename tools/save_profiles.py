"""Copy what a gpurun profiling call left under gpurun_out/ into profiles/ (tracked): the rows of
rocprofv3's kernel stats for this library's kernels and the runtime's copy / fill kernels as they are,
the torch kernels that only build the synthetic input summed into one row; the bench lines; the HBM
traffic of the scan from the PMC passes (tools/traffic.sh).   usage: tools/save_profiles.py <tag> <name>"""
import csv, json, sys
from pathlib import Path
R = Path(__file__).resolve().parent.parent
tag, name = sys.argv[1], sys.argv[2]
def stats(src, dst):
    rows = list(csv.reader(open(src)))
    head, body = rows[0], rows[1:]
    keep = [r for r in body if 'adsb::' in r[0] or '__amd_rocclr' in r[0]]
    other = [r for r in body if r not in keep]
    tot = sum(int(r[2]) for r in body)
    if other:
        keep.append(['torch kernels that build the synthetic input (summed)', str(sum(int(r[1]) for r in other)),
                     str(sum(int(r[2]) for r in other)), '', f"{100.0 * sum(int(r[2]) for r in other) / tot:.2f}", '', '', ''])
    with open(dst, 'w', newline='') as f:
        w = csv.writer(f); w.writerow(head[:8]); w.writerows([r[:8] for r in keep])
stats(R / f'gpurun_out/prof_{tag}/bench_kernel_stats.csv', R / f'profiles/{name}_kernel_stats.csv')
stats(R / f'gpurun_out/psync_{tag}/bench_kernel_stats.csv', R / f'profiles/{name}_sync_kernel_stats.csv')
for src, dst in ((f'gpurun_out/bench_{tag}.json', f'profiles/{name}_bench.json'), (f'gpurun_out/bench_sync_{tag}.json', f'profiles/{name}_sync_bench.json')):
    line = [l for l in open(R / src).read().splitlines() if l.startswith('{')][-1]
    (R / dst).write_text(line + '\n')
for w in ('dense', 'stream', 'shard', 'live', 'gloo2', 'gloo2_shard', 'gloo8', 'gloo8_shard'):  # tools/profile_all.sh
    src = R / f'gpurun_out/bench_{w}_{tag}.json'
    if src.exists():
        line = [l for l in src.read_text().splitlines() if l.startswith('{')][-1]
        (R / f'profiles/{name}_{w}_bench.json').write_text(line + '\n')
for src, dst in ((f'gpurun_out/hosttime_ring_{tag}.txt', f'profiles/{name}_hosttime_ring.txt'), (f'gpurun_out/config1_{tag}.txt', f'profiles/{name}_config1.txt')):
    if (R / src).exists():
        (R / dst).write_text(''.join(l for l in open(R / src) if 'amdgpu.ids' not in l))
if (R / f'gpurun_out/pdense_{tag}/bench_kernel_stats.csv').exists():
    stats(R / f'gpurun_out/pdense_{tag}/bench_kernel_stats.csv', R / f'profiles/{name}_dense_sync_kernel_stats.csv')
tr = json.loads([l for l in open(R / f'gpurun_out/traffic_{tag}.json').read().splitlines() if l.startswith('{')][-1])
lib = json.loads((R / f'profiles/{name}_bench.json').read_text())['config']['library']
fetch, write = tr['scan_FETCH_SIZE_avg'] * 1024 * 2, tr['scan_WRITE_SIZE_avg'] * 1024
out = {"library": lib, "chunks": 512, "bytes_per_launch": int(fetch + write), "fetch_bytes": int(fetch), "write_bytes": int(write),
       "algorithmic_bytes": 268435456, "raw": tr,
       "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py (tools/traffic.sh); "
                 "FETCH_SIZE x 1024 x 2 (gfx950 counts wide coalesced reads at half, MI355X_MICROARCH.md HBM section), "
                 "WRITE_SIZE x 1024 uncalibrated"}
(R / 'profiles/scan_hbm_traffic.json').write_text(json.dumps(out, indent=1) + '\n')
print(out['bytes_per_launch'], lib)
sq = R / f'gpurun_out/sq_{tag}/sq_counters.json'   # tools/sq_counters.py sq_<tag>
if sq.exists():
    (R / f'profiles/{name}_sq_counters.json').write_text(sq.read_text())
    (R / 'profiles/scan_sq_counters.json').write_text(sq.read_text())
    print('sq counters', json.loads(sq.read_text()).get('valu_roofline', {}).get('frac'))
