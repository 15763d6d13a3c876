#!/bin/bash
# round 6, visit 19: wave-slot priority in the exact-f32 fused inverted-residual kernel: CamVid-L (levels 3-5 on it), M / S level 3
tag=${1:-r6v19}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp HS_K1_CHAIN=1 HS_IR_MATH=auto
out=$R/gpurun_out/irf_prio_$tag.txt; : > $out
for rep in 1 2; do
for cfg in Lc M S; do
for v in product irf_prio_young irf_prio_old; do
  lib=$R/hyperseg_amd/lib/libhyperseg_hip_$v.so; [ $v = product ] && lib=$R/hyperseg_amd/lib/libhyperseg_hip.so
  rm -rf /tmp/pv; cd /tmp
  HS_HIP_LIB=$lib timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -- python $R/tools/decoder_loop.py $cfg 40 > /tmp/pv.log 2>&1
  cd $R; f=$(find /tmp/pv -name '*kernel_stats.csv' | head -1)
  echo "== $cfg $v rep=$rep | $(grep 'graph replay' /tmp/pv.log)" | tee -a $out
  [ -n "$f" ] && python tools/kstats.py $f patch_ir_fused 4 | cut -c1-140 | tee -a $out
done; done; done
