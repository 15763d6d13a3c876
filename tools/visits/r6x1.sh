#!/bin/bash
# round 6, visit x1: BatchNorm's two-launch kernels with a batch of loads in flight per thread (was: one load per iteration, a full round trip
# each): the training tests, the graphed step's time, the EXACT per-step launch table (fp32)
tag=${1:-r6x1}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_training.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/tests_train_$tag.txt
timeout 200 python tools/train_step_time.py 50 graph graph_bf16 2>&1 | tail -2 | tee gpurun_out/train_step_$tag.txt
for n in 10 30; do
  ( cd /tmp && rm -rf /tmp/prof_t_$n && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t_$n -- python $R/tools/train_step_time.py $n fp32 > /tmp/prof_t_$n.log 2>&1
    f=$(find /tmp/prof_t_$n -name '*kernel_stats.csv' | head -1); cp "$f" $R/gpurun_out/train_stats_fp32_${n}_$tag.csv )
done
python tools/train_launch_count.py gpurun_out/train_stats_fp32_10_$tag.csv 10 gpurun_out/train_stats_fp32_30_$tag.csv 30 60 | cut -c1-170 | tee gpurun_out/train_launches_$tag.txt
