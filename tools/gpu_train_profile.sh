#!/bin/bash
# Per-kernel picture of the config-5 decoder training step (CamVid-S 576x576 bs 2) under rocprofv3, fp32 and bf16 autocast.
#   gpurun --timeout 600 -- 'bash tools/gpu_train_profile.sh <tag>'
tag=${1:-x}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python tools/train_step_time.py 20 > gpurun_out/train_step_$tag.txt 2>&1; cat gpurun_out/train_step_$tag.txt | tail -3
for mode in fp32 bf16; do
  ( cd /tmp && rm -rf /tmp/prof_train_$mode && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train_$mode -- python $R/tools/train_step_time.py 20 $mode > /tmp/prof_train_$mode.log 2>&1
    f=$(find /tmp/prof_train_$mode -name '*kernel_stats.csv' | head -1)
    if [ -n "$f" ]; then cp "$f" $R/gpurun_out/train_kernel_stats_${mode}_$tag.csv; echo "== $mode (22 steps incl. 2 warm-up + model setup)"; python $R/tools/kstats.py "$f" "" 22; else echo "no stats $mode"; tail -5 /tmp/prof_train_$mode.log; fi )
done
python tools/graph_floor.py 200 2>&1 | tail -3 | tee gpurun_out/graph_floor_$tag.txt
