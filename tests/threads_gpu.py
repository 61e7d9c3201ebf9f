"""Two host threads, each with its own context on the same GPU, demodulating different captures at
the same time (contexts are independent streams; a context itself is not thread-safe).  Checks
every result against the oracle.  Test infrastructure (uses oracle/)."""
import sys, threading
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
from dump1090_rs_amd import Context, synth
from oracle import binding

def key(m): return (m.chunk, m.j, m.try_phase, m.score, m.msg, m.signal_level)
def okey(w): return (w["chunk"], w["j"], w["try_phase"], w["score"], w["msg"], w["signal_level"])
errors = []
def work(tid):
    try:
        n = (24 + 8 * tid) * 131072 - 1000 * tid
        iq = synth.make_iq(n, n_bursts=300, seed=900 + tid, n_icao=20, df11_every=4)
        want = [okey(w) for w in binding.Oracle().demod_iq(iq)[0]]
        dev = torch.from_numpy(iq).cuda()
        ctx = Context(0, 24 + 8 * tid)
        for it in range(40):
            ctx.icao_flush()
            if it % 2:
                got = [key(m) for m in ctx.demod_iq_device(dev.data_ptr(), n)]
            else:  # pipelined: the same capture twice
                ctx.submit_iq_device(dev.data_ptr(), n); ctx.icao_flush(); ctx.submit_iq_device(dev.data_ptr(), n)
                got = [key(m) for m in ctx.collect()]
                if [key(m) for m in ctx.collect()] != want: errors.append((tid, it, "second"))
            if got != want: errors.append((tid, it, "first"))
        ctx.close()
    except Exception as e:  # noqa
        errors.append((tid, repr(e)))
ts = [threading.Thread(target=work, args=(k,)) for k in range(3)]
[t.start() for t in ts]; [t.join() for t in ts]
print("errors:", errors)
sys.exit(1 if errors else 0)
