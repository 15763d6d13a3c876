#!/bin/bash
# round 6, visit x24: the bf16 step after its own upsample2x kernel and the batched staging of plain_fwd_kernel: the whole GPU suite's training +
# parity files that touch them, step times, bf16 launch list
tag=${1:-r6x24}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_training.py tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/tests_$tag.txt
timeout 200 python tools/train_step_time.py 50 graph graph_bf16 2>&1 | tail -2 | tee gpurun_out/train_step_$tag.txt
( cd /tmp && rm -rf /tmp/prof_tg && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tg -- python $R/tools/prof_train_graph.py 20 bf16 > /tmp/prof_tg.log 2>&1
  f=$(find /tmp/prof_tg -name '*kernel_trace.csv' | head -1)
  if [ -n "$f" ]; then python $R/tools/frame_sequence.py "$f" 20 | cut -c1-230 > $R/gpurun_out/train_sequence_bf16_$tag.txt; else tail -20 /tmp/prof_tg.log; fi )
tail -1 gpurun_out/train_sequence_bf16_$tag.txt
