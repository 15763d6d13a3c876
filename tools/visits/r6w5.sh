#!/bin/bash
# round 6, visit w5: lean fused expand + depthwise kernel with stores left in flight across the chunk loop: tests, then tile / chunk-group sweep
#   gpurun --timeout 1200 -- 'bash tools/visits/r6w5.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp HS_BENCH_MBCONV_FIRST_TABLE_ONLY=1
out=$R/gpurun_out/mbx_lean_r6w5.txt; : > $out
timeout 300 python -m pytest tests/test_hip_encoder.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -5 | tee -a $out
run() { echo "== $*" | tee -a $out; env "$@" timeout 120 python tools/bench_mbconv.py 2>&1 | tail -12 | grep -E "^ +[2-8] |sum" | cut -c1-75 | tee -a $out; }
run HS_MBX_LEAN=0
for mw in 768 512 384 256 128; do
  run HS_MBX_LEAN=1 HS_MBX_MIN_WG=$mw
  run HS_MBX_LEAN=1 HS_MBX_MIN_WG=$mw HS_MBX_OTH1=8 HS_MBX_OTH2=4
done
run HS_MBX_LEAN=1 HS_MBX_MIN_WG=1024 HS_MBX_OTH1=8 HS_MBX_OTH2=4
run HS_MBX_LEAN=1 HS_MBX_MIN_WG=1536 HS_MBX_OTH1=8 HS_MBX_OTH2=4
