#!/bin/bash
# round 6, visit 17: which form is faster for CamVid HyperSeg-L's narrow inverted residuals -- exact-f32 fused kernel vs f16-split kernel (what hs_ir_math AUTO should pick)
tag=${1:-r6v17}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp HS_K1_CHAIN=1
out=$R/gpurun_out/lc_math_$tag.txt; : > $out
for rep in 1 2; do
for cfg in Lc Sc; do
for mode in f32 auto; do
  rm -rf /tmp/pv; cd /tmp
  HS_IR_MATH=$mode timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -- python $R/tools/decoder_loop.py $cfg 40 > /tmp/pv.log 2>&1
  cd $R; f=$(find /tmp/pv -name '*kernel_stats.csv' | head -1)
  echo "== $cfg $mode rep=$rep | $(grep 'graph replay' /tmp/pv.log)" | tee -a $out
  [ -n "$f" ] && python tools/kstats.py $f patch_ir 6 | cut -c1-140 | tee -a $out
done; done; done
