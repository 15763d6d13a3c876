#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
for v in "" _nostore _ntstore; do
  rm -rf /tmp/prof_L$v
  ( cd /tmp && HS_HIP_LIB=$R/hyperseg_amd/lib/libhyperseg_hip$v.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_L$v -- python $R/tools/decoder_loop.py L 10 > /tmp/prof_L$v.log 2>&1 )
  echo "== variant '$v'"; grep -E "decoder graph" /tmp/prof_L$v.log
  f=$(find /tmp/prof_L$v -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && python tools/kstats.py "$f" "patch_ir" 40
done
