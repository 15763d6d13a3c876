#!/bin/bash
# Exact per-frame launch table of the benched configuration: two profiled bench runs with different step counts (set-up cancels).
tag=${1:-r4s}; shift; extra="$@"; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for n in 50 150; do
  ( cd /tmp && rm -rf /tmp/prof_f_$n && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_f_$n -- python $R/bench.py --steps $n --warmup 10 --repeats 1 --no-extras --no-cpu-baseline --traffic off $extra > /tmp/prof_f_$n.log 2>&1
    tail -1 /tmp/prof_f_$n.log | cut -c1-200
    f=$(find /tmp/prof_f_$n -name '*kernel_stats.csv' | head -1); cp "$f" $R/gpurun_out/frame_stats_${n}_$tag.csv )
done
python tools/train_launch_count.py gpurun_out/frame_stats_50_$tag.csv 50 gpurun_out/frame_stats_150_$tag.csv 150 80 | cut -c1-200 | tee gpurun_out/frame_launches_$tag.txt
