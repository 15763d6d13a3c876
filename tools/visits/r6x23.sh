#!/bin/bash
# round 6, visit x23: in-order launch list of one replayed config-5 step under bf16 autocast, beside the fp32 one
tag=${1:-r6x23}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for mode in fp32 bf16; do
( cd /tmp && rm -rf /tmp/prof_tg && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tg -- python $R/tools/prof_train_graph.py 20 $mode > /tmp/prof_tg.log 2>&1
  f=$(find /tmp/prof_tg -name '*kernel_trace.csv' | head -1)
  if [ -n "$f" ]; then python $R/tools/frame_sequence.py "$f" 20 | cut -c1-230 > $R/gpurun_out/train_sequence_${mode}_$tag.txt; else tail -20 /tmp/prof_tg.log; fi )
tail -1 gpurun_out/train_sequence_${mode}_$tag.txt
done
