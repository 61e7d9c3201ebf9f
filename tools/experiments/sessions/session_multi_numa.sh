#!/bin/bash
# Round 5: does it matter where the threads of a dense adsb_multi capture run?  (tuning build: stage times of the parallel replay)
cd ${GRAFT_REPO_ROOT:-.}
cp dump1090_rs_amd/libadsb_hip.so /tmp/rel.so
trap 'cp /tmp/rel.so dump1090_rs_amd/libadsb_hip.so' EXIT
cp variants/lib_tune.so dump1090_rs_amd/libadsb_hip.so
BUS=$(python - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print(f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0")
PY
)
NODE=$(cat /sys/bus/pci/devices/$BUS/numa_node 2>/dev/null); LIST=$(cat /sys/devices/system/node/node$NODE/cpulist 2>/dev/null)
echo "gpu $BUS node $NODE cpus $LIST; nproc $(nproc); cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
for how in "" "taskset -c $LIST"; do for n in 1 8; do
  echo "== contexts $n, dense, ${how:-unpinned}"
  ADSB_HOST_TIMES=1 $how python tools/multi_steps.py --contexts $n --chunks 4096 --steps 12 --pipelined --bursts 5000 2>&1 | grep "device thread 0:\|parallel replay\|^{" | cut -c1-420
done; done
