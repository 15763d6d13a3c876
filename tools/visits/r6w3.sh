#!/bin/bash
# round 6, visit w3: the lean fused expand + depthwise kernel (hs_mbconv_lean.hip): encoder tests, per-block times lean on / off, tile knobs
#   gpurun --timeout 900 -- 'bash tools/visits/r6w3.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp HS_BENCH_MBCONV_FIRST_TABLE_ONLY=1
out=$R/gpurun_out/mbx_lean_r6w3.txt; : > $out
timeout 300 python -m pytest tests/test_hip_encoder.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -5 | tee -a $out
run() { echo "== $*" | tee -a $out; env "$@" timeout 120 python tools/bench_mbconv.py 2>&1 | tail -12 | grep -E "^ +[2-8] |sum" | tee -a $out; }
run HS_MBX_LEAN=0
run HS_MBX_LEAN=1
run HS_MBX_LEAN=1 HS_MBX_OTH1=8
run HS_MBX_LEAN=1 HS_MBX_OTH2=4
run HS_MBX_LEAN=1 HS_MBX_OTH1=8 HS_MBX_OTH2=4 HS_MBX_MIN_WG=1536
run HS_MBX_LEAN=1 HS_MBX_MIN_WG=512
run HS_MBX_LEAN=1 HS_MBX_MIN_WG=1536
