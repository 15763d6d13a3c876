#!/bin/bash
# Round-4 visit f: new tests (general MetaConv2d backward, co-scheduling) + the config-5 training step's per-kernel picture.
tag=${1:-r4f}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "meta_conv2d or coscheduled or validation_after or nan or foreign or checkpoint or in_graph" 2>&1 | tail -5
timeout 200 python tools/train_step_time.py 20 > gpurun_out/train_step_$tag.txt 2>&1; cat gpurun_out/train_step_$tag.txt | tail -4
for mode in fp32 bf16; do
  ( cd /tmp && rm -rf /tmp/prof_train_$mode && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train_$mode -- python $R/tools/train_step_time.py 20 $mode > /tmp/prof_train_$mode.log 2>&1
    f=$(find /tmp/prof_train_$mode -name '*kernel_stats.csv' | head -1)
    if [ -n "$f" ]; then cp "$f" $R/gpurun_out/train_kernel_stats_${mode}_$tag.csv; echo "== $mode (22 steps incl. 2 warm-up + model setup)"; python $R/tools/kstats.py "$f" "" 90 | cut -c1-150; else echo "no stats $mode"; tail -5 /tmp/prof_train_$mode.log; fi )
done
