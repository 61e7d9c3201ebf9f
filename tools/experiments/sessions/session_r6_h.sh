#!/bin/bash
# Round 6 session H: the final tree once more -- the GPU suite, smoke, and two more soak seeds.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6_h
mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest gpu rc=$?"; tail -3 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
for seed in 31337 8675309; do
  ( time timeout 1500 python tests/fuzz_gpu.py --cases 2000 --seed $seed --dense 50 --mixed 200 --multi 700 ) > $O/soak_$seed.txt 2>&1; echo "soak $seed rc=$?"; grep -v amdgpu.ids $O/soak_$seed.txt | head -1 | cut -c1-1400
done
