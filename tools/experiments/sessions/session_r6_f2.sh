#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6_f
mkdir -p $O
ASAN=$(gcc -print-file-name=libasan.so)
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=0
echo "== import torch under the preload"; LD_PRELOAD=$ASAN timeout 300 python -X faulthandler -c "import torch; print(torch.cuda.is_available()); x = torch.zeros(10, device='cuda'); print(x.sum().item())" > $O/torch_preload.log 2>&1; echo "rc=$?"; grep -v "^    #" $O/torch_preload.log | head -30 | cut -c1-300; grep -m1 -A14 "ERROR: AddressSanitizer" $O/torch_preload.log | cut -c1-200
