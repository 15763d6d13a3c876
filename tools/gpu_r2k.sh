#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
tag=${1:-r2k}
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/prof_train && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -- python $R/tools/train_step_time.py 10 > /tmp/prof_train.log 2>&1
  grep "training step" /tmp/prof_train.log
  f=$(find /tmp/prof_train -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then cp "$f" $R/gpurun_out/train_kernel_stats_$tag.csv; python $R/tools/kstats.py "$f" "" 30; else echo "no stats"; tail -5 /tmp/prof_train.log; fi )
