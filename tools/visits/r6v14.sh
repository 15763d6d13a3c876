#!/bin/bash
# round 6, visit 14: wave-slot priority in the chain kernel (younger / older workgroup of a CU first), two interleaved repetitions
tag=${1:-r6v14}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp HS_IR_MATH=auto HS_K1_CHAIN=1
out=$R/gpurun_out/kc_prio_$tag.txt; : > $out
for rep in 1 2; do
for cfg in M Sc S Lc; do
for v in product kc_prio_young kc_prio_old; do
  lib=$R/hyperseg_amd/lib/libhyperseg_hip_$v.so; [ $v = product ] && lib=$R/hyperseg_amd/lib/libhyperseg_hip.so
  rm -rf /tmp/pv; cd /tmp
  HS_HIP_LIB=$lib timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -- python $R/tools/decoder_loop.py $cfg 40 > /tmp/pv.log 2>&1
  cd $R; f=$(find /tmp/pv -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && echo "$cfg $v rep=$rep $(python tools/kstats.py $f chain 2 | tr '\n' ' ' | cut -c1-140) | $(grep 'graph replay' /tmp/pv.log)" | tee -a $out
done; done; done
