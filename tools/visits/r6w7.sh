#!/bin/bash
# round 6, visit w7: output staging (64-byte store requests) in the lean fused expand + depthwise kernel: encoder tests, per-block times
# against the kernel without it (variant library mbl_nostage), whole-frame A/B
#   gpurun --timeout 1200 -- 'bash tools/visits/r6w7.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp HS_BENCH_MBCONV_FIRST_TABLE_ONLY=1
out=$R/gpurun_out/mbx_stage_r6w7.txt; : > $out
timeout 300 python -m pytest tests/test_hip_encoder.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -5 | tee -a $out
for v in mbl_nostage product; do
  lib=$R/hyperseg_amd/lib/libhyperseg_hip_$v.so; [ $v = product ] && lib=$R/hyperseg_amd/lib/libhyperseg_hip.so
  for knobs in "HS_MBX_MIN_WG=768" "HS_MBX_MIN_WG=512" "HS_MBX_MIN_WG=768 HS_MBX_OTH1=8"; do
    echo "== $v $knobs" | tee -a $out
    env $knobs HS_HIP_LIB=$lib timeout 120 python tools/bench_mbconv.py 2>&1 | tail -12 | grep -E "^ +[2-8] |sum" | cut -c1-75 | tee -a $out
  done
done
for round in 1 2; do
  for v in mbl_nostage product; do
    lib=$R/hyperseg_amd/lib/libhyperseg_hip_$v.so; [ $v = product ] && lib=$R/hyperseg_amd/lib/libhyperseg_hip.so
    HS_HIP_LIB=$lib timeout 200 python bench.py --model m --steps 300 --warmup 30 --no-cpu-baseline --no-extras 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$round m $v', d['value'], d['ms_per_step'])" | tee -a $out
  done
done
