#!/bin/bash
# Round-3 visit for the Op C kernel (hs_patch_irc.hip): the inverted-residual parity tests in all math modes, then per-kernel
# times of the decoder loop (rocprofv3 --kernel-trace --stats) with the f16-split form on.
#   gpurun --timeout 300 -- 'bash tools/gpu_irc.sh <tag> [configs...]'
tag=${1:-x}; shift; cfgs=${@:-M}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
out=$R/gpurun_out/irc_$tag.txt; : > $out
timeout 200 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "${IRC_TESTS:-inverted_residual or split_ir or full_config or tiny_decoder or modes_agree or op_c}" 2>&1 | tail -15 | tee -a $out
for c in $cfgs; do
  rm -rf /tmp/prof_$c; cd /tmp
  HS_IR_MATH=auto timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$c -- python $R/tools/decoder_loop.py $c 40 > /tmp/prof_$c.log 2>&1
  cd $R
  tail -2 /tmp/prof_$c.log | tee -a $out
  f=$(find /tmp/prof_$c -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then cp $f gpurun_out/irc_${tag}_${c}_kernel_stats.csv; echo "== $c" | tee -a $out; python tools/kstats.py $f hs:: 60 | tee -a $out; fi
done
