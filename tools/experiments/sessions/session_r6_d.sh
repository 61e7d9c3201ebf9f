#!/bin/bash
# Round 6 session D: everything profiles/r6_v21_* holds (tools/profile.sh), the GPU suite on the final library, and the round's
# soak: every entry point + dense / mixed pipelines + adsb_multi sequences with injected shard failures, restarts and wait modes.
cd ${GRAFT_REPO_ROOT:-.}
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_r6v21.log 2>&1; echo "pytest gpu rc=$?"; tail -3 gpurun_out/pytest_gpu_r6v21.log
./tools/profile.sh r6v21 2>&1 | grep -v amdgpu.ids | cut -c1-400
( time timeout 1500 python tests/fuzz_gpu.py --cases 1500 --seed 606 --dense 40 --mixed 150 --multi 600 ) > gpurun_out/soak_r6v21.txt 2>&1; echo "soak rc=$?"; grep -v amdgpu.ids gpurun_out/soak_r6v21.txt | tail -12
