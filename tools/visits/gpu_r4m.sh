#!/bin/bash
# Round-4 visit m: exact per-step launch count of the config-5 training step (two profiled runs with different step counts), step times.
tag=${1:-r4m}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python -m pytest tests -m gpu -q -p no:cacheprovider -x tests/test_hip_training.py -k "not config5_full" 2>&1 | tail -3
timeout 200 python tools/train_step_time.py 20 2>&1 | grep -v Warn | tail -4 | tee gpurun_out/train_step_$tag.txt
echo "== with the signal's gradient"; HS_TRAIN_SIGNAL_GRAD=1 timeout 200 python tools/train_step_time.py 20 2>&1 | grep -v Warn | tail -4 | tee -a gpurun_out/train_step_$tag.txt
for mode in fp32 sg; do
  [ $mode = sg ] && export HS_TRAIN_SIGNAL_GRAD=1
  m2=fp32
  for n in 10 30; do
    ( cd /tmp && rm -rf /tmp/prof_t_${mode}_$n && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t_${mode}_$n -- python $R/tools/train_step_time.py $n $m2 > /tmp/prof_t_${mode}_$n.log 2>&1
      f=$(find /tmp/prof_t_${mode}_$n -name '*kernel_stats.csv' | head -1); cp "$f" $R/gpurun_out/train_stats_${mode}_${n}_$tag.csv )
  done
  echo "== $mode"; python tools/train_launch_count.py gpurun_out/train_stats_${mode}_10_$tag.csv 10 gpurun_out/train_stats_${mode}_30_$tag.csv 30 16 | cut -c1-170
done | tee gpurun_out/train_launches_$tag.txt
