#!/bin/bash
# round 6, visit w9: stem + block 0's depthwise half as one launch (hs_stem_dw_fwd): encoder / model tests, whole-frame A/B (HS_STEM_DW=0|1),
# and the output-staging A/B of the lean kernel done properly (variant library mbl_nostage = -DHS_MBL_STAGE=0), interleaved twice
#   gpurun --timeout 1500 -- 'bash tools/visits/r6w9.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp HS_BENCH_MBCONV_FIRST_TABLE_ONLY=1
out=$R/gpurun_out/stem_dw_r6w9$1.txt; : > $out
if [ "$1" != "ab" ]; then
timeout 600 python -m pytest tests/test_hip_encoder.py tests/test_model_boundary.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -5 | tee -a $out
for v in product mbl_nostage; do
  lib=$R/hyperseg_amd/lib/libhyperseg_hip_$v.so; [ $v = product ] && lib=$R/hyperseg_amd/lib/libhyperseg_hip.so
  echo "== $v" | tee -a $out
  HS_HIP_LIB=$lib timeout 120 python tools/bench_mbconv.py 2>&1 | tail -12 | grep -E "^ +[2-8] |sum" | cut -c1-75 | tee -a $out
done
fi
for round in 1 2; do
  for m in m s; do
    for cfg in "product 0" "product 1" "mbl_nostage 1"; do
      set -- $cfg; v=$1; sd=$2
      lib=$R/hyperseg_amd/lib/libhyperseg_hip_$v.so; [ $v = product ] && lib=$R/hyperseg_amd/lib/libhyperseg_hip.so
      HS_STEM_DW=$sd HS_HIP_LIB=$lib timeout 200 python bench.py --model $m --steps 300 --warmup 30 --no-cpu-baseline --no-extras 2>/dev/null | \
        python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$round $m $v stem_dw=$sd', d['value'], d['ms_per_step'])" | tee -a $out
    done
  done
done
