"""tools/experiments/sessions/session_ablate.sh's output -> profiles/scan_stage_split.json: the vector wave-instructions k_scan_fast issues per launch in
each stage (SQ_INSTS_VALU with the kernel cut short after the stage, ADSB_DEBUG_STOP, differences), per wave-tile, and the blocking
launch durations of the cuts -- the P1-only one is the memory floor of this access pattern (the whole HBM read, no later stage).
usage: python tools/stage_split.py gpurun_out/r5_ablate.txt "<library version string>" > profiles/scan_stage_split.json"""
import json, re, sys
txt = open(sys.argv[1]).read()
valu, us = {}, {}
for m in re.finditer(r"stop=(\d+) (.*?) us=([\d.]+)", txt):
    kv = dict(x.split("=") for x in m.group(2).split())
    valu[int(m.group(1))] = float(kv["SQ_INSTS_VALU"])
    us[int(m.group(1))] = float(m.group(3))
blocking = {int(a): float(b) * 1e3 for a, b, _ in re.findall(r"^(\d) ([\d.]+) ([\d.]+)$", txt, flags=re.M)}
order = [(1, "P1 magnitudes (+ tables into LDS)"), (2, "P2 sign planes"), (3, "P3 patterns (+ per-lane counts, prefix scan)"),
         (6, "compaction of the matches"), (4, "P4 value gates"), (5, "P5 trials"), (0, "tile epilogue")]
wave_tiles = 512 * 17 * 4
stages, prev = [], 0.0
for stop, name in order:
    d = valu[stop] - prev
    stages.append({"stage": name, "cut": stop, "valu_wave_insts": int(d), "per_wave_tile": round(d / wave_tiles, 1),
                   "share": round(d / valu[0], 4), "us_blocking_cut_after": round(blocking.get(stop, 0.0), 1), "us_under_counters": us[stop]})
    prev = valu[stop]
out = {"library": sys.argv[2], "chunks": 512, "wave_tiles_per_launch": wave_tiles, "valu_wave_insts_per_launch": int(valu[0]), "stages": stages,
       "memory_floor_ms": round(blocking[1] / 1e3, 4),
       "memory_floor_is": "k_scan_fast cut after P1 (ADSB_DEBUG_STOP=1, tuning build, blocking launches, tools/ablate.py --sync): the whole "
                          "HBM read of a launch with none of the later stages -- what this access pattern (8080 magnitudes per 7712 positions, "
                          "eight 16-byte loads per thread in flight, four workgroups per CU) takes at best",
       "source": "tools/experiments/sessions/session_ablate.sh: rocprofv3 --pmc SQ_INSTS_VALU ... over bench.py with ADSB_DEBUG_STOP = 1, 2, 3, 6, 4, 5, 0 (differences between "
                 "consecutive cuts), and tools/ablate.py --sync for the durations"}
print(json.dumps(out, indent=1))
