#!/bin/bash
# round 5, visit 19: EXACT per-step launch table of the config-5 training step (fp32) on the round's final training code
tag=${1:-r5v19}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for n in 10 30; do
  ( cd /tmp && rm -rf /tmp/prof_t_$n && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t_$n -- python $R/tools/train_step_time.py $n fp32 > /tmp/prof_t_$n.log 2>&1
    f=$(find /tmp/prof_t_$n -name '*kernel_stats.csv' | head -1); cp "$f" $R/gpurun_out/train_stats_fp32_${n}_$tag.csv )
done
python tools/train_launch_count.py gpurun_out/train_stats_fp32_10_$tag.csv 10 gpurun_out/train_stats_fp32_30_$tag.csv 30 60 | cut -c1-170 | tee gpurun_out/train_launches_$tag.txt
