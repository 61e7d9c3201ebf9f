import sys, time, ctypes as C
sys.path.insert(0,'.')
import torch
from dump1090_rs_amd import Context, synth
from dump1090_rs_amd._lib import AdsbMsg
n=512*131072
bufs=[synth.make_iq_torch(n, n_bursts=64, seed=synth.SEED_DEFAULT+b, device='cuda') for b in range(3)]
torch.cuda.synchronize()
ctx=Context(0,512); cap=1<<20; out=(AdsbMsg*cap)()
def run(label, flush=True, stats=True, prof=True, steps=30):
    ctx.set_profiling(prof)
    for i in range(3):
        ctx.icao_flush(); ctx.demod_iq_device_raw(bufs[i%3].data_ptr(), n, out, cap)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for i in range(steps):
        if flush: ctx.icao_flush()
        ctx.demod_iq_device_raw(bufs[i%3].data_ptr(), n, out, cap)
        if stats: ctx.stats()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/steps*1e6
    print(f"{label:40s} {dt:8.1f} us/step")
run("flush+demod+stats, profiling on")
run("flush+demod, profiling on", stats=False)
run("flush+demod, profiling off", stats=False, prof=False)
run("demod only, profiling off", flush=False, stats=False, prof=False)
ctx.set_profiling(True); ctx.icao_flush(); ctx.demod_iq_device_raw(bufs[0].data_ptr(), n, out, cap); print(ctx.stats())
