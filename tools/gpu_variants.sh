#!/bin/bash
# Per-kernel decoder times (rocprofv3 --kernel-trace --stats) of the product library and of variant libraries.
#   gpurun --timeout 400 -- 'bash tools/gpu_variants.sh <tag> <cfg> [variant names...]'
tag=${1:-x}; cfg=${2:-M}; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp HS_IR_MATH=auto
out=$R/gpurun_out/variants_$tag.txt; : > $out
for v in product "$@"; do
  lib=$R/hyperseg_amd/lib/libhyperseg_hip_$v.so; [ $v = product ] && lib=$R/hyperseg_amd/lib/libhyperseg_hip.so
  [ -f $lib ] || { echo "missing $lib" | tee -a $out; continue; }
  rm -rf /tmp/pv; cd /tmp
  HS_HIP_LIB=$lib timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -- python $R/tools/decoder_loop.py $cfg 40 > /tmp/pv.log 2>&1
  cd $R; echo "== $v ($cfg)" | tee -a $out
  f=$(find /tmp/pv -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && python tools/kstats.py $f hs:: 60 | tee -a $out
done
