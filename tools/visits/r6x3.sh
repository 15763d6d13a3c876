#!/bin/bash
# round 6, visits x3, x4: the training step's data-movement kernels with more work per thread (dw_tiles_bwd_in two tile rows, dw_tiles_bwd_w a channel
# per 16-lane row for 8 x 8 patches, halo_tiles_bwd 2 x 2 candidates in flight, halo_tiles_fwd / stage_input_plane four rows, upsample2x_bwd
# 2 x 2 pixels from pair loads): training tests, replayed step time, in-order launch list
tag=${1:-r6x3}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_training.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/tests_train_$tag.txt
timeout 200 python tools/train_step_time.py 50 graph graph_bf16 2>&1 | tail -2 | tee gpurun_out/train_step_$tag.txt
( cd /tmp && rm -rf /tmp/prof_tg && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tg -- python $R/tools/prof_train_graph.py 20 > /tmp/prof_tg.log 2>&1
  f=$(find /tmp/prof_tg -name '*kernel_trace.csv' | head -1)
  if [ -n "$f" ]; then python $R/tools/frame_sequence.py "$f" 20 | cut -c1-230 > $R/gpurun_out/train_sequence_$tag.txt; else tail -20 /tmp/prof_tg.log; fi )
tail -1 gpurun_out/train_sequence_$tag.txt
