#!/bin/bash
# round 5, visit 10: the wide one-launch SE gate for the early blocks (A/B against a -DHS_SE_FUSED_WIDE=0 build), SE tails opt-in tests
tag=${1:-r5v10}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_encoder.py -m gpu -q -p no:cacheprovider > gpurun_out/pytest_encoder_$tag.log 2>&1
echo "encoder pytest rc=$?"; tail -4 gpurun_out/pytest_encoder_$tag.log | cut -c1-300
grep -E "^E  " gpurun_out/pytest_encoder_$tag.log | head -30 | cut -c1-400
bash tools/gpu_ab_env.sh sefw_$tag "HS_SE_WIDE=default" "HS_HIP_LIB=$R/hyperseg_amd/lib/libhyperseg_hip_sefw0.so"
