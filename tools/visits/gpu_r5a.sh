#!/bin/bash
# training visit: tests, step times, exact launch table (fp32), dW slice-size variants of the s2w backward
tag=${1:-r5a}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x tests/test_hip_training.py 2>&1 | tail -4 | tee gpurun_out/pytest_train_$tag.log
timeout 200 python tools/train_step_time.py 20 2>&1 | grep -v Warn | tail -4 | tee gpurun_out/train_step_$tag.txt
for n in 10 30; do
  ( cd /tmp && rm -rf /tmp/prof_t_$n && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t_$n -- python $R/tools/train_step_time.py $n fp32 > /tmp/prof_t_$n.log 2>&1
    f=$(find /tmp/prof_t_$n -name '*kernel_stats.csv' | head -1); cp "$f" $R/gpurun_out/train_stats_fp32_${n}_$tag.csv )
done
python tools/train_launch_count.py gpurun_out/train_stats_fp32_10_$tag.csv 10 gpurun_out/train_stats_fp32_30_$tag.csv 30 34 | cut -c1-170 | tee gpurun_out/train_launches_$tag.txt
for v in ; do
  export HS_HIP_LIB=$R/hyperseg_amd/lib/libhyperseg_hip_$v.so
  ( cd /tmp && rm -rf /tmp/prof_v && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_v -- python $R/tools/train_step_time.py 10 fp32 > /tmp/prof_v.log 2>&1
    f=$(find /tmp/prof_v -name '*kernel_stats.csv' | head -1); echo "$v: $(grep -E 's2w_train_bwd_kernel<0>|s2w_train_dwsum' $f | awk -F'"' '{split($3,q,","); printf "%s avg %.2f us | ", substr($2,1,40), q[4]/1000}')" )
done | tee gpurun_out/st_slice_$tag.txt
