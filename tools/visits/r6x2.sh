#!/bin/bash
# round 6, visit x2: in-order launch list of ONE replayed config-5 training step (durations, gaps, grids)
tag=${1:-r6x2}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/prof_tg && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tg -- python $R/tools/prof_train_graph.py 20 > /tmp/prof_tg.log 2>&1
  f=$(find /tmp/prof_tg -name '*kernel_trace.csv' | head -1)
  if [ -n "$f" ]; then python $R/tools/frame_sequence.py "$f" 20 | cut -c1-230 > $R/gpurun_out/train_sequence_$tag.txt; else tail -20 /tmp/prof_tg.log; fi )
tail -3 gpurun_out/train_sequence_$tag.txt
