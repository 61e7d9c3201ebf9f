#!/bin/bash
# End-to-end rate of adsb_feed (capture file -> `*hex;` lines) over a 1 GiB capture in /dev/shm.
# usage: gpurun -- ./tools/feed_rate.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
python - <<'PY'
import torch, numpy as np
from dump1090_rs_amd import synth
n = 2048 * 131072 // 8                      # 1 GiB / 8 per piece
with open("/dev/shm/adsb_cap.iq", "wb") as f:
    for k in range(8):
        t = synth.make_iq_torch(n, n_bursts=64 * 256 // 512 * 8, seed=1000 + k, device="cuda")
        f.write(t.cpu().numpy().tobytes())
print("wrote 1 GiB")
PY
for rep in 1 2; do
for r in 1 2 4 8; do echo -n "--buffers 64 --readers $r: "; ./dump1090_rs_amd/adsb_feed --readers $r --buffers 64 /dev/shm/adsb_cap.iq 2>&1 > /dev/null | grep "^adsb_feed:"; done
for b in 1 4 16 64; do
  for order in "--mem-order" ""; do
    echo -n "--buffers $b ${order:-(file order: swapped on the way in)}: "
    ./dump1090_rs_amd/adsb_feed $order --buffers $b /dev/shm/adsb_cap.iq 2>&1 > /dev/null | grep "^adsb_feed:"
  done
done
done
rm -f /dev/shm/adsb_cap.iq
