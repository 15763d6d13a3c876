#!/bin/bash
# final-state bf16 launch table of the config-5 training step (two profiled runs, set-up cancels)
tag=${1:-r5j}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for n in 10 30; do
  ( cd /tmp && rm -rf /tmp/prof_t_$n && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t_$n -- python $R/tools/train_step_time.py $n bf16 > /tmp/prof_t_$n.log 2>&1
    f=$(find /tmp/prof_t_$n -name '*kernel_stats.csv' | head -1); cp "$f" $R/gpurun_out/train_stats_bf16_${n}_$tag.csv )
done
python tools/train_launch_count.py gpurun_out/train_stats_bf16_10_$tag.csv 10 gpurun_out/train_stats_bf16_30_$tag.csv 30 60 | cut -c1-170 | tee gpurun_out/train_launches_bf16_$tag.txt
