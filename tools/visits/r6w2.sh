#!/bin/bash
# round 6, visit w2: tile shape / chunk-group count of the fused expand + depthwise launch (hs_mbconv.hip), per block (tools/bench_mbconv.py)
#   gpurun --timeout 900 -- 'bash tools/visits/r6w2.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp HS_BENCH_MBCONV_FIRST_TABLE_ONLY=1
out=$R/gpurun_out/mbx_tiles_r6w2.txt; : > $out
run() { echo "== $*" | tee -a $out; env "$@" timeout 120 python tools/bench_mbconv.py 2>&1 | grep -v "^ *[0-9]* *[0-9]* *[0-9]* [35] [12] .* nan" | tail -12 | tee -a $out; }
run HS_MBX_OTH1=16 HS_MBX_OTH2=8 HS_MBX_MIN_WG=768
for mw in 256 512 1024 1536; do
  run HS_MBX_OTH1=16 HS_MBX_OTH2=8 HS_MBX_MIN_WG=$mw
done
for mw in 256 512 768 1024 1536; do
  run HS_MBX_OTH1=8 HS_MBX_OTH2=4 HS_MBX_MIN_WG=$mw
done
run HS_MBX_OTH1=8 HS_MBX_OTH2=8 HS_MBX_MIN_WG=512
run HS_MBX_OTH1=16 HS_MBX_OTH2=4 HS_MBX_MIN_WG=512
