#!/bin/bash
# round 6, visit w13: the lean fused expand + depthwise kernel at Cin = 80 (KS = 20; blocks 9-12 of HyperSeg-M, 64 x 32 maps) against the GEMM +
# depthwise pair those blocks run: per block (tools/bench_mbconv.py), then whole frame with the fusion threshold raised (HS_FUSE_EXPAND_MAX_CIN=80)
#   gpurun --timeout 1200 -- 'bash tools/visits/r6w13.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp HS_BENCH_MBCONV_FIRST_TABLE_ONLY=1
out=$R/gpurun_out/lean_ks20_r6w13.txt; : > $out
timeout 300 python -m pytest tests/test_hip_encoder.py -m gpu -q -p no:cacheprovider -x -k "mbconv" 2>&1 | tail -3 | tee -a $out
run() { echo "== $*" | tee -a $out; env "$@" timeout 120 python tools/bench_mbconv.py 2>&1 | tail -12 | grep -E "^ +(9|1[0-2]) |sum" | cut -c1-75 | tee -a $out; }
run HS_MBX_LEAN=0
run HS_MBX_LEAN=1
run HS_MBX_LEAN=1 HS_MBX_OTH1=8
run HS_MBX_LEAN=1 HS_MBX_OTH1=8 HS_MBX_MIN_WG=256
for round in 1 2; do
  for cfg in "40 16" "80 16" "80 8"; do
    set -- $cfg
    HS_FUSE_EXPAND_MAX_CIN=$1 HS_MBX_OTH1=$2 timeout 200 python bench.py --model m --steps 300 --warmup 30 --no-cpu-baseline --no-extras 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$round m max_cin=$1 oth1=$2', d['value'], d['ms_per_step'])" | tee -a $out
  done
done
