#!/bin/bash
# round 6, visit w4: phase removal (timing-only variant libraries) in the lean fused expand + depthwise kernel, per block
#   gpurun --timeout 900 -- 'bash tools/visits/r6w4.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp HS_BENCH_MBCONV_FIRST_TABLE_ONLY=1
out=$R/gpurun_out/mbx_lean_phases_r6w4.txt; : > $out
for v in product mbl_noswish mbl_nostore mbl_noload mbl_nomfma mbl_nodw; do
  lib=$R/hyperseg_amd/lib/libhyperseg_hip_$v.so; [ $v = product ] && lib=$R/hyperseg_amd/lib/libhyperseg_hip.so
  echo "== $v" | tee -a $out
  HS_HIP_LIB=$lib timeout 120 python tools/bench_mbconv.py 2>&1 | tail -12 | grep -E "^ +[2-8] |sum" | tee -a $out
done
