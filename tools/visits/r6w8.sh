#!/bin/bash
# round 6, visit w8: in-order dispatch list of one graph-replayed HyperSeg-M frame (durations, gaps, grids) on the current library
#   gpurun --timeout 600 -- 'bash tools/visits/r6w8.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp && rm -rf /tmp/pf && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pf -- python $R/tools/prof_graph.py 20 fold_dw > /tmp/pf.log 2>&1
f=$(find /tmp/pf -name '*kernel_trace.csv' | head -1); cd $R
python tools/frame_sequence.py $f 20 > gpurun_out/frame_sequence_r6w8.txt 2>&1; tail -3 gpurun_out/frame_sequence_r6w8.txt
