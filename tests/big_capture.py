"""One-off: a capture far larger than the bench's (default 2048 buffers = 1 GiB of IQ) through
adsb_demod_iq_device against the multi-threaded oracle.  Test infrastructure (uses oracle/)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
from dump1090_rs_amd import Context, synth
from oracle import binding

chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n = chunks * 131072 - 12345
dev = synth.make_iq_torch(n, n_bursts=40 * chunks // 8, seed=4711, n_icao=300, df11_every=6, device="cuda")
torch.cuda.synchronize()
host = dev.cpu().numpy()
t = time.time(); want, _ = binding.Oracle().demod_iq(host, cap=1 << 22, threads=64); t_cpu = time.time() - t
ctx = Context(0, chunks)
ctx.icao_flush()
t = time.time(); got = ctx.demod_iq_device(dev.data_ptr(), n, cap=1 << 22); t_gpu = time.time() - t
a = [(m.chunk, m.j, m.try_phase, m.score, m.msg, m.signal_level) for m in got]
b = [(w["chunk"], w["j"], w["try_phase"], w["score"], w["msg"], w["signal_level"]) for w in want]
from dump1090_rs_amd._lib import AdsbMsg
out = (AdsbMsg * (1 << 20))()
for rep in range(3):  # without the Python-side unpacking of the frame list
    ctx.icao_flush(); t = time.time(); k = ctx.demod_iq_device_raw(dev.data_ptr(), n, out, 1 << 20)
    print(f"  raw call {rep}: {(time.time() - t) * 1e3:.2f} ms, {k} frames")
    # (from the second call on the context knows the stream is dense: ordered -- k_order_prefix, more
    #  than 1024 buffers -- and scored on the device)
    again = [(m.chunk, m.j, m.try_phase, m.score, bytes(m.msg), m.signal_level) for m in out[:k]]
    if again != b:
        print("  MISMATCH in raw call", rep)
        sys.exit(1)
print(f"{chunks} buffers, {n} samples: {len(b)} frames, identical={a == b}, oracle (64 threads) {t_cpu:.2f} s, GPU call {t_gpu * 1e3:.1f} ms, stats {ctx.stats()}")
sys.exit(0 if a == b else 1)
