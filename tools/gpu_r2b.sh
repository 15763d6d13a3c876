#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
tag=${1:-r2b}
timeout 600 python -m pytest tests/test_hip_training.py -m gpu -q -p no:cacheprovider -k "config5 or hyperseg_m_level" > gpurun_out/pytest_train_$tag.log 2>&1
grep -E "tensor-relative|passed|failed" gpurun_out/pytest_train_$tag.log | cut -c1-3000
HS_HIP_LIB=$R/hyperseg_amd/lib/libhyperseg_hip_stamps.so timeout 300 python tools/ir_phase_times.py M > gpurun_out/phases_${tag}_M.txt 2>&1; cat gpurun_out/phases_${tag}_M.txt | tail -40
HS_HIP_LIB=$R/hyperseg_amd/lib/libhyperseg_hip_stamps.so timeout 300 python tools/ir_phase_times.py L > gpurun_out/phases_${tag}_L.txt 2>&1; cat gpurun_out/phases_${tag}_L.txt | tail -40
bash tools/pmc_decoder.sh $tag 2>&1 | grep -A30 "patch_ir_fused_kernel<34" | head -100
