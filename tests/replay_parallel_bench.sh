#!/bin/bash
# usage (GPU box or here; CPU only): tests/replay_parallel_bench.sh [copies]   -- per-record cost of the serial replay and of the
# parallel replay's scan / score stages on one thread (what the pool divides)
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
python - <<'PY'
import ctypes as C, numpy as np, sys
sys.path.insert(0, '.')
from oracle import binding
from dump1090_rs_amd import synth
from dump1090_rs_amd._lib import AdsbTrial
O = binding.lib()
n_buf = 64
iq = synth.make_iq(n_buf * 131072, n_bursts=5000 * n_buf // 512, seed=99, n_icao=200)
recs = []
for c, off in enumerate(range(0, len(iq), 131072)):
    mb = binding.OrcMagBuf()
    O.orc_to_mag(np.ascontiguousarray(iq[off:off + 131072]).ctypes.data, 131072, C.byref(mb))
    buf = (AdsbTrial * (5 * 131072 // 8))()
    n = O.orc_all_trials(C.byref(mb), c, buf, len(buf))
    recs += [bytes(buf[i]) for i in range(n)]
open('/tmp/recs.bin', 'wb').write(b"".join(recs))
PY
g++ -O3 -std=c++17 -pthread tests/replay_parallel_bench.cpp dump1090_rs_amd/csrc/adsb_replay_host.cpp -o /tmp/replay_parallel_bench && /tmp/replay_parallel_bench ${1:-64}
