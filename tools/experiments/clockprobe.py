"""Does a blocking one-buffer call get slower when the host dawdles between calls (GPU clocks follow
utilisation)?  Same call, same data, a busy-wait of D microseconds after each; prints the call's own time."""
import sys, time, ctypes as C
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
from dump1090_rs_amd import Context, synth
from dump1090_rs_amd._lib import AdsbMsg

raw = np.fromfile(ROOT / "tests" / "golden" / "test_1641427457780.iq", dtype="<i2").reshape(-1, 2)
fix = np.ascontiguousarray(raw[:, ::-1])
syn = synth.make_iq(131072, n_bursts=1, seed=5)
ctx = Context(0, 1)
L, h = ctx._L, ctx._h
out = (AdsbMsg * 4096)(); nn = C.c_size_t()
for name, iq in (("fixture", fix), ("synthetic", syn)):
    dev = torch.from_numpy(iq).cuda(); ptr = C.c_void_p(dev.data_ptr()); n = len(iq)
    for api in ("blocking", "submit+collect"):
        for delay_us in (0, 10, 30, 100):
            def call():
                if api == "blocking":
                    L.adsb_demod_iq_device(h, ptr, n, out, 4096, C.byref(nn))
                else:
                    L.adsb_submit_iq_device(h, ptr, n); L.adsb_collect(h, out, 4096, C.byref(nn))
            t_end = time.perf_counter() + 0.3
            while time.perf_counter() < t_end: call()
            spent, reps = 0.0, 0
            t_end = time.perf_counter() + 0.7
            while time.perf_counter() < t_end:
                a = time.perf_counter(); call(); b = time.perf_counter()
                spent += b - a; reps += 1
                while time.perf_counter() - b < delay_us * 1e-6: pass
            print(f"{name:9s} {api:15s} +{delay_us:3d} us idle between calls: {spent / reps * 1e6:6.1f} us per call", flush=True)
