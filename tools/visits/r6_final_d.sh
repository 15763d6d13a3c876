#!/bin/bash
# round 6, evidence visit D (after the training-step work of the round's last session): everything of visit A (r6_final_a.sh) + the config-5
# step's EXACT launch table (two rocprofv3 runs, set-up cancels) and the in-order launch list of one replayed step.
tag=${1:-r6gc}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/visits/r6_final_a.sh $tag
timeout 200 python tools/train_step_time.py 50 graph graph_bf16 2>&1 | tail -2 | tee gpurun_out/train_step_$tag.txt
for n in 10 30; do
  ( cd /tmp && rm -rf /tmp/prof_t_$n && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t_$n -- python $R/tools/train_step_time.py $n fp32 > /tmp/prof_t_$n.log 2>&1
    f=$(find /tmp/prof_t_$n -name '*kernel_stats.csv' | head -1); cp "$f" $R/gpurun_out/train_stats_fp32_${n}_$tag.csv )
done
python tools/train_launch_count.py gpurun_out/train_stats_fp32_10_$tag.csv 10 gpurun_out/train_stats_fp32_30_$tag.csv 30 60 | cut -c1-170 > gpurun_out/train_launches_$tag.txt; head -3 gpurun_out/train_launches_$tag.txt
( cd /tmp && rm -rf /tmp/prof_tg && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_tg -- python $R/tools/prof_train_graph.py 20 > /tmp/prof_tg.log 2>&1
  f=$(find /tmp/prof_tg -name '*kernel_trace.csv' | head -1)
  if [ -n "$f" ]; then python $R/tools/frame_sequence.py "$f" 20 | cut -c1-230 > $R/gpurun_out/train_sequence_$tag.txt; else tail -20 /tmp/prof_tg.log; fi )
tail -1 gpurun_out/train_sequence_$tag.txt
