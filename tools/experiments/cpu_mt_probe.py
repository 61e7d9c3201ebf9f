"""How the N-thread oracle (the bench's cpu_baseline.all_cores) scales on this box: seconds inside the C call."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from oracle import binding
from dump1090_rs_amd import synth
native = binding.build_native()
L = binding.load(native) if native else None
iq = synth.make_iq(512 * 131072, n_bursts=64, seed=synth.SEED_DEFAULT)
big = np.ascontiguousarray(np.tile(iq, (4, 1)))
for name, arr in (("256 MiB", iq), ("1 GiB", big)):
    for t in (1, 32, 64, 128, 256):
        if t == 1 and arr is big: continue
        o = binding.Oracle(L); tm = []
        for _ in range(3 if t > 1 else 1):
            o.icao_flush(); o.demod_iq(arr, cap=1 << 20, threads=t, timing=tm)
        s = sorted(tm)[len(tm) // 2]
        print(f"{name:8s} {t:4d} threads: {s * 1e3:8.1f} ms  {len(arr) / s / 1e6:9.1f} Msamples/s", flush=True)
