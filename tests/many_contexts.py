import sys, time; sys.path.insert(0, '.')
import numpy as np, torch
from dump1090_rs_amd import Context
raw = np.fromfile("tests/golden/test_1641427457780.iq", dtype="<i2").reshape(-1, 2); iq = np.ascontiguousarray(raw[:, ::-1])
dev = torch.from_numpy(iq).cuda()
t = time.time(); ctxs = [Context(0, 1) for _ in range(48)]; print("created 48 contexts in %.2f s" % (time.time() - t))
free, total = torch.cuda.mem_get_info(); print("device memory in use: %.1f GiB" % ((total - free) / 2**30))
ok = True
for rep in range(5):
    for c in ctxs: c.icao_flush(); c.submit_iq_device(dev.data_ptr(), len(iq))
    for c in ctxs: ok &= len(c.collect()) == 5
t = time.time()
for rep in range(20):
    for c in ctxs: c.submit_iq_device(dev.data_ptr(), len(iq))
    for c in ctxs: ok &= len(c.collect()) >= 5
dt = time.time() - t
print("all ok:", ok, "; 48 streams x 20 buffers in %.1f ms = %.2f Gsample/s" % (dt * 1e3, 48 * 20 * 131072 / dt / 1e9))
for c in ctxs: c.close()
