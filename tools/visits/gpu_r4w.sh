#!/bin/bash
# Round-4 visit w: phase-removal variants of s2w_train_bwd_kernel<0> (timing only) + the reworked tiny-patch input gradient.
tag=${1:-r4w}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider -x tests/test_hip_training.py -k "gradients_vs_oracle or bf16_storage or train_step" 2>&1 | tail -3 | tee gpurun_out/pytest_train_$tag.log
timeout 200 python tools/train_step_time.py 20 graph graph_bf16 2>&1 | grep -v Warn | tail -2 | tee gpurun_out/train_step_$tag.txt
for v in product s2wt_noA s2wt_noB s2wt_nomfma s2wt_nostore; do
  [ $v = product ] && unset HS_HIP_LIB || export HS_HIP_LIB=$R/hyperseg_amd/lib/libhyperseg_hip_$v.so
  ( cd /tmp && rm -rf /tmp/prof_v && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_v -- python $R/tools/train_step_time.py 10 fp32 > /tmp/prof_v.log 2>&1
    f=$(find /tmp/prof_v -name '*kernel_stats.csv' | head -1)
    echo "$v: $(grep -E 's2w_train_bwd_kernel<0>|patch_conv_bwd_input_tiny' $f | awk -F, '{printf "%s calls %s avg %.2f us | ", substr($1,1,60), $2, $4/1000}')" )
done | tee gpurun_out/s2wt_probe_$tag.txt
