#!/bin/bash
# quick visit: s2w training tests, replayed step time, per-kernel times of the s2w backward
tag=${1:-r4z}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider -x tests/test_hip_training.py -k "s2w or train_step or batchnorm" 2>&1 | tail -3 | tee gpurun_out/pytest_train_$tag.log
timeout 200 python tools/train_step_time.py 20 graph graph_bf16 2>&1 | grep -v Warn | tail -2 | tee gpurun_out/train_step_$tag.txt
HS_TRAIN_SIGNAL_GRAD=1 timeout 200 python tools/train_step_time.py 20 graph 2>&1 | grep -v Warn | tail -1 | tee -a gpurun_out/train_step_$tag.txt
( cd /tmp && rm -rf /tmp/prof_v && HS_TRAIN_SIGNAL_GRAD=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_v -- python $R/tools/train_step_time.py 10 fp32 > /tmp/prof_v.log 2>&1
  f=$(find /tmp/prof_v -name '*kernel_stats.csv' | head -1); grep -E 's2w_train' $f | awk -F, '{printf "%s calls %s avg %.2f us\n", substr($1,1,70), $2, $4/1000}' ) | tee gpurun_out/s2wt_$tag.txt
