#!/usr/bin/env python3
"""SQ / GRBM counter passes over bench.py for k_scan_fast, raw rows kept.

    python3 tools/sq_counters.py <tag> [--pipelined]      (on the GPU box, from the repo root)

Runs rocprofv3 --pmc in separate passes (8 SQ slots per pass on gfx950, MI355X_MICROARCH.md
"rocprofv3 PMC slots"), each with --kernel-trace only, over

    python3 bench.py --sync --steps 20 --warmup 3 --no-cpu-baseline --no-also      (default)

(blocking calls: launches do not overlap, a launch's counters are the kernel alone) and writes
gpurun_out/<tag>/sq_counters.json: per-launch averages of every counter over the k_scan_fast
dispatches of the 256 MiB workload, the derived VALU roofline, and the raw per-dispatch rows of
the first launches (so that the averages can be recomputed from the file).  Copy it to profiles/.

SQ_WAVE_CYCLES / SQ_BUSY_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (4 shader
clocks); SQ_INSTS_* count wave-instructions.  This process never touches the GPU itself: it only
starts rocprofv3 as a child.
"""
from __future__ import annotations

import collections
import csv
import glob
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(os.environ.get("GRAFT_REPO_ROOT") or Path(__file__).resolve().parent.parent)
PASSES = [
    ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
     "SQ_INSTS_VALU", "SQ_INSTS_LDS"],
    ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_SALU",
     "SQ_INSTS_VMEM_RD", "SQ_WAIT_INST_LDS", "GRBM_GUI_ACTIVE"],
    ["SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC",
     "SQ_THREAD_CYCLES_VALU", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_UNALIGNED_STALL"],
]
N_CU, SIMD_PER_CU, N_XCC, PEAK_GHZ = 256, 4, 8, 2.4


def run_pass(tag: str, k: int, counters, bench_args):
    out = ROOT / "gpurun_out" / tag / f"pass{k}"
    out.mkdir(parents=True, exist_ok=True)
    cmd = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", str(out), "-o", "p",
           "--", "python3", str(ROOT / "bench.py"), *bench_args]
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    files = glob.glob(str(out / "**" / "p_counter_collection.csv"), recursive=True)
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    line = None
    for ln in r.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = json.loads(ln)
    return rows, line, r.returncode


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "sq"
    pipelined = "--pipelined" in sys.argv
    bench_args = ["--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-also"]
    if not pipelined:
        bench_args.insert(0, "--sync")
    per_dispatch = collections.defaultdict(dict)  # (pass, dispatch id) -> counter -> value
    grid, dur_ns = {}, {}
    library = None
    lines = []
    for k, counters in enumerate(PASSES):
        rows, line, rc = run_pass(tag, k, counters, bench_args)
        lines.append({"pass": k, "rc": rc, "ms_per_step": line and line.get("ms_per_step"),
                      "kernel_avg_ms": line and line["roofline"].get("kernel_avg_ms")})
        if line:
            library = line["config"]["library"]
        for r in rows:
            if "k_scan_fast" not in r["Kernel_Name"]:
                continue
            did = (k, int(r["Dispatch_Id"]))
            per_dispatch[did][r["Counter_Name"]] = per_dispatch[did].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            grid[did] = int(r.get("Grid_Size", 0) or 0)
            dur_ns[did] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    # keep the launches of the 256 MiB workload only: the full persistent grid (1024 workgroups x 256)
    full = max(grid.values()) if grid else 0
    agg = collections.defaultdict(list)
    raw = []
    for did in sorted(per_dispatch):
        if grid[did] != full:
            continue
        for name, v in per_dispatch[did].items():
            agg[name].append(v)
        if len(raw) < 3 * 24:
            raw.append({"pass": did[0], "dispatch": did[1], **{k: int(v) for k, v in per_dispatch[did].items()}})
    avg = {k: sum(v) / len(v) for k, v in agg.items()}
    n = {k: len(v) for k, v in agg.items()}
    out = {"library": library, "chunks": 512,
           "what": "rocprofv3 --pmc passes over `python3 bench.py " + " ".join(bench_args) + "`, per-launch averages for "
                   "adsb::k_scan_fast<false> (full-grid launches of the 256 MiB workload only)",
           "units": "SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* : quad-cycles summed over all waves (or SIMDs for BUSY); "
                    "SQ_INSTS_* : wave-instructions; GRBM_GUI_ACTIVE: shader clocks the GPU was busy",
           "passes": PASSES, "bench_lines": lines, "launches_averaged": n, "per_launch_avg": {k: round(v, 1) for k, v in avg.items()},
           "grid_size": full}
    durs = [dur_ns[d] for d in per_dispatch if grid[d] == full]
    try:
        valu = avg["SQ_INSTS_VALU"]
        act = avg["SQ_ACTIVE_INST_VALU"] * 4.0          # shader clocks the vector pipes were busy, summed over SIMDs
        dur = sum(durs) / len(durs)                      # ns, the dispatch's own begin/end stamps under rocprofv3
        clocks_xcc = avg.get("GRBM_GUI_ACTIVE", 0.0) / N_XCC
        simd_cycles = N_CU * SIMD_PER_CU * dur * PEAK_GHZ
        out["valu_roofline"] = {
            "wave_insts": round(valu), "busy_clocks_per_inst": round(act / valu, 3),
            "launch_ns_under_pmc": round(dur, 1),
            "simd_cycles_available": round(simd_cycles), "valu_busy_clocks": round(act),
            "frac": round(act / simd_cycles, 4),
            "is": "SQ_ACTIVE_INST_VALU x 4 clocks / (256 CUs x 4 SIMDs x launch duration x 2.4 GHz peak shader clock): the "
                  "share of all SIMD clocks of the launch in which a vector instruction was executing (a launch on its own "
                  "includes its ramp and its half-empty last round of tiles; the counter resolves quad-cycles, so an "
                  "instruction that issues in under 4 clocks still counts 4)",
            "grbm_gui_active_per_xcc": round(clocks_xcc), "grbm_over_launch_GHz": round(clocks_xcc / dur, 3) if dur else None,
            "wave_cycle_split": {
                "issuing": round(avg["SQ_ACTIVE_INST_ANY"] / avg["SQ_WAVE_CYCLES"], 4),
                "stalled_at_issue": round(avg["SQ_WAIT_INST_ANY"] / avg["SQ_WAVE_CYCLES"], 4),
                "parked_waitcnt_or_barrier": round(avg["SQ_WAIT_ANY"] / avg["SQ_WAVE_CYCLES"], 4),
                "valu_share_of_wave_cycles": round(avg["SQ_ACTIVE_INST_VALU"] / avg["SQ_WAVE_CYCLES"], 4)},
        }
    except (KeyError, ZeroDivisionError) as e:
        out["valu_roofline_error"] = repr(e)
    dst = ROOT / "gpurun_out" / tag / "sq_counters.json"
    # (the raw rows one per line: the file stays readable and diffable)
    head = json.dumps(out, indent=1)
    rows = ",\n".join("  " + json.dumps(r, separators=(",", ":")) for r in raw)
    dst.write_text(head[:-2] + ',\n "raw_rows_first_launches": [\n' + rows + "\n ]\n}\n")
    print(json.dumps({k: out[k] for k in ("per_launch_avg", "valu_roofline", "bench_lines") if k in out}, indent=1))


if __name__ == "__main__":
    main()
