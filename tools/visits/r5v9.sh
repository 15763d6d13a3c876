#!/bin/bash
# round 5, visit 9: per-dispatch durations of one replayed HyperSeg-M frame with and without the SE tail (which blocks gain, which lose)
tag=${1:-r5v9}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in 1 0; do
  rm -rf /tmp/pv; cd /tmp
  HS_SE_TAIL=$v timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pv -- python $R/tools/prof_graph.py 40 dw > /tmp/pv.log 2>&1
  cd $R
  f=$(find /tmp/pv -name '*kernel_trace.csv' | head -1)
  [ -n "$f" ] && python tools/frame_sequence.py $f 40 > gpurun_out/frame_sequence_se${v}_$tag.txt
  tail -1 gpurun_out/frame_sequence_se${v}_$tag.txt
done
