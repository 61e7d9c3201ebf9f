"""BASELINE config 1: the reference's `cargo bench` case (benches/demod_benchmark.rs:7-12) --
icao_flush + to_mag + demodulate2400 on test_iq/test_1641427457780.iq (131072 samples) -- timed
through the drop-in entry points.  Prints microseconds per iteration."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
from dump1090_rs_amd import Context

raw = np.fromfile(ROOT / "tests" / "golden" / "test_1641427457780.iq", dtype="<i2").reshape(-1, 2)
iq = np.ascontiguousarray(raw[:, ::-1])           # file order is [im][re]
ctx = Context(0, 1)
dev = torch.from_numpy(iq).cuda()

def timeit(fn, seconds=1.0):
    """mean over at least `seconds` of back-to-back calls after 0.3 s of them untimed (criterion warms up
    for 3 s and measures for 5: a burst of 200 calls after idling times the GPU at its idle clocks)"""
    t = time.perf_counter()
    while time.perf_counter() - t < 0.3: fn()
    torch.cuda.synchronize(); t = time.perf_counter(); n = 0
    while time.perf_counter() - t < seconds:
        for _ in range(50): fn()
        n += 50
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e6

def ref_api():
    ctx.icao_flush(); m = ctx.to_mag(iq); return ctx.demodulate2400(m)
def fused_host():
    ctx.icao_flush(); return ctx.demod_iq(iq)
def fused_dev():
    ctx.icao_flush(); return ctx.demod_iq_device(dev.data_ptr(), len(iq))
assert len(ref_api()) == len(fused_host()) == len(fused_dev()) == 5
print(f"to_mag + demodulate2400 (host buffers, the reference's API shape): {timeit(ref_api):8.1f} us")
print(f"adsb_demod_iq (host IQ in, fused):                                {timeit(fused_host):8.1f} us")
print(f"adsb_demod_iq_device (IQ resident):                               {timeit(fused_dev):8.1f} us")
# the ABI calls alone (no Python list of messages built per call)
from dump1090_rs_amd._lib import AdsbMsg
import ctypes as C
out = (AdsbMsg * 4096)(); nn = C.c_size_t(); L = ctx._L; h = ctx._h
def raw_host():
    L.adsb_icao_flush(h); L.adsb_demod_iq(h, iq.ctypes.data, len(iq), out, 4096, C.byref(nn))
def raw_dev():
    L.adsb_icao_flush(h); L.adsb_demod_iq_device(h, C.c_void_p(dev.data_ptr()), len(iq), out, 4096, C.byref(nn))
def raw_dev_noflush():
    L.adsb_demod_iq_device(h, C.c_void_p(dev.data_ptr()), len(iq), out, 4096, C.byref(nn))
print(f"  raw ABI: icao_flush + adsb_demod_iq (host IQ):                  {timeit(raw_host):8.1f} us")
print(f"  raw ABI: icao_flush + adsb_demod_iq_device:                     {timeit(raw_dev):8.1f} us")
print(f"  raw ABI: adsb_demod_iq_device, no flush:                        {timeit(raw_dev_noflush):8.1f} us")
ctx.set_profiling(0)
print(f"  ... the same three with HIP-event timing off (adsb_set_profiling 0): {timeit(raw_host):6.1f} / {timeit(raw_dev):6.1f} / {timeit(raw_dev_noflush):6.1f} us")
ctx.set_profiling(1)
m = ctx.to_mag(iq)
print(f"  adsb_to_mag alone:                                              {timeit(lambda: ctx.to_mag(iq)):8.1f} us")
print(f"  adsb_demodulate2400 alone:                                      {timeit(lambda: (ctx.icao_flush(), ctx.demodulate2400(m))):8.1f} us")
