#!/bin/bash
# round 6, visit w15: does the lean fused expand + depthwise launch gain from FEWER co-resident workgroups (more generations: loads of one
# generation under the compute / stores of another)?  HS_MBX_LDS_PAD pads the dynamic LDS segment (43 KB + pad KB per workgroup).
#   gpurun --timeout 900 -- 'bash tools/visits/r6w15.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp HS_BENCH_MBCONV_FIRST_TABLE_ONLY=1
out=$R/gpurun_out/lean_occupancy_r6w15.txt; : > $out
run() { echo "== $*" | tee -a $out; env "$@" timeout 120 python tools/bench_mbconv.py 2>&1 | tail -12 | grep -E "^ +[2-8] |sum" | cut -c1-75 | tee -a $out; }
run HS_MBX_LDS_PAD=0
run HS_MBX_LDS_PAD=10
run HS_MBX_LDS_PAD=20
run HS_MBX_LDS_PAD=20 HS_MBX_MIN_WG=1536
run HS_MBX_LDS_PAD=20 HS_MBX_OTH1=8 HS_MBX_OTH2=4 HS_MBX_MIN_WG=1536
run HS_MBX_LDS_PAD=0 HS_MBX_OTH1=8 HS_MBX_OTH2=4 HS_MBX_MIN_WG=3000
run HS_MBX_LDS_PAD=20 HS_MBX_OTH1=8 HS_MBX_OTH2=4 HS_MBX_MIN_WG=3000
