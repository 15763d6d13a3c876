#!/bin/bash
# round 5, visit 17: one-launch Adam + the shared bank-gradient buffer: training tests, step times with torch's fused Adam vs ours
tag=${1:-r5v17}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_training.py tests/test_model_boundary.py -m gpu -q -p no:cacheprovider > gpurun_out/pytest_training_$tag.log 2>&1
echo "training pytest rc=$?"; tail -4 gpurun_out/pytest_training_$tag.log | cut -c1-300
grep -E "^E  " gpurun_out/pytest_training_$tag.log | head -30 | cut -c1-400
for adam in torch ours; do
  echo "-- HS_ADAM=$adam" | tee -a gpurun_out/train_step_$tag.txt
  HS_ADAM=$adam timeout 200 python tools/train_step_time.py 30 graph graph_bf16 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/train_step_$tag.txt
done
echo "-- HS_ADAM=ours, USE_SHARED_BANK_GRAD off" | tee -a gpurun_out/train_step_$tag.txt
HS_ADAM=ours HS_SHARED_BANK_GRAD=0 timeout 200 python tools/train_step_time.py 30 graph graph_bf16 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/train_step_$tag.txt
