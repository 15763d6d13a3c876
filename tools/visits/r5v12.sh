#!/bin/bash
# round 5, visit 12: BatchNorm2 on load in the inverted residual's last 1x1 layer (autograd.PatchConvBN): training tests, step time, launch count
tag=${1:-r5v12}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_training.py tests/test_model_boundary.py -m gpu -q -p no:cacheprovider > gpurun_out/pytest_training_$tag.log 2>&1
echo "training pytest rc=$?"; tail -6 gpurun_out/pytest_training_$tag.log | cut -c1-300
grep -E "^E  " gpurun_out/pytest_training_$tag.log | head -30 | cut -c1-400
timeout 200 python tools/train_step_time.py 30 graph graph_bf16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/train_step_$tag.txt
HS_X=1 timeout 200 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/train_step_$tag.txt
import hyperseg_amd.autograd as HA
HA.USE_CONV_BN_FUSED = False
import runpy, sys
sys.argv = ['tools/train_step_time.py', '30', 'graph', 'graph_bf16']
print('-- USE_CONV_BN_FUSED = False')
runpy.run_path('tools/train_step_time.py', run_name='__main__')
PY
