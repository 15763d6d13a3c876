#!/bin/bash
# round 6, visit 6: timing-only bound of a barrier-free chunk loop for the level-4 kernel; per-kernel decoder tables S / L / Lc with the
# 16-byte-store signal2weights.
tag=${1:-r6v6}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for cfg in M S; do
  bash tools/gpu_variants.sh ${tag}_$cfg $cfg irc_nobarrier > /dev/null 2>&1
  cat gpurun_out/variants_${tag}_$cfg.txt | grep -E "==|patch_irc" | cut -c1-160
done
export HS_IR_MATH=auto
for cfg in S Sc L Lc; do
  rm -rf /tmp/pv_$cfg; cd /tmp
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv_$cfg -- python $R/tools/decoder_loop.py $cfg 30 > /tmp/pv.log 2>&1
  cd $R; echo "== $cfg" | tee -a gpurun_out/decoder_kernels_$tag.txt; grep decoder /tmp/pv.log | tee -a gpurun_out/decoder_kernels_$tag.txt
  f=$(find /tmp/pv_$cfg -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && python tools/kstats.py $f hs:: 12 | cut -c1-150 | tee -a gpurun_out/decoder_kernels_$tag.txt
done
