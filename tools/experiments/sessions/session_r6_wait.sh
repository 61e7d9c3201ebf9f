#!/bin/bash
# Round 6, VERDICT r5 item 5: adsb_multi's wait policy under 4 CPUs, 16 CPUs and no affinity limit -> profiles/r6_wait_policy.txt
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6_wait.txt
: > $O
for cpus in 0-3 0-15 all; do
  echo "## affinity: $cpus" >> $O
  if [ $cpus = all ]; then timeout 600 python tools/wait_policy.py >> $O 2>&1; else timeout 600 taskset -c $cpus python tools/wait_policy.py >> $O 2>&1; fi
done
grep -v amdgpu.ids $O
