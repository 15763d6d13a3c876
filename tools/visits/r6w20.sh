#!/bin/bash
# round 6, visit w20: what runs in a HyperSeg-L bs 32 step -- rocprofv3 kernel stats of `bench.py --model l --no-extras` at two step counts
# (the difference cancels set-up, parity pass and capture)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for n in 20 60; do
  ( cd /tmp && rm -rf /tmp/prof_l_$n && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_l_$n -- python $R/bench.py --model l --steps $n --warmup 5 --repeats 1 --no-extras --no-cpu-baseline --traffic off > /tmp/prof_l_$n.log 2>&1
    tail -1 /tmp/prof_l_$n.log | cut -c1-160
    f=$(find /tmp/prof_l_$n -name '*kernel_stats.csv' | head -1); cp "$f" $R/gpurun_out/l_stats_${n}_r6w20.csv )
done
python tools/train_launch_count.py gpurun_out/l_stats_20_r6w20.csv 20 gpurun_out/l_stats_60_r6w20.csv 60 60 | cut -c1-200 | tee gpurun_out/l_launches_r6w20.txt
