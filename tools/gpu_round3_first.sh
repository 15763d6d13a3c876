#!/bin/bash
# First GPU visit of round 3 (everything here was prepared, compile-checked and CPU-tested at the end of round 2, none of it has
# run inside the model on a GPU).  Build the variants on the CPU first:  python tools/build_variants.py kpreload dwtile_k5 dwtile_k5_128
#   gpurun --timeout 420 -- 'bash tools/gpu_round3_first.sh <tag>'
tag=${1:-r5a}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/round3_first_$tag.txt; : > $out
line() { python bench.py --no-extras --no-cpu-baseline --steps 200 --warmup 20 --repeats 3 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
echo "== opt-in split-GEMM tests" | tee -a $out
HS_TEST_SPLIT_GEMM=1 timeout 200 python -m pytest tests/test_split_gemm.py -q -p no:cacheprovider 2>&1 | tail -4 | tee -a $out
echo "== bench M: default | --split-gemm" | tee -a $out
line | tee -a $out
line --split-gemm | tee -a $out
for v in kpreload dwtile_k5 dwtile_k5_128; do
  lib=$R/hyperseg_amd/lib/libhyperseg_hip_$v.so
  if [ -f $lib ]; then
    echo "== variant $v: encoder tests, bench M" | tee -a $out
    HS_HIP_LIB=$lib timeout 120 python -m pytest tests/test_hip_encoder.py -q -p no:cacheprovider -x 2>&1 | tail -2 | tee -a $out
    HS_HIP_LIB=$lib line | tee -a $out
  fi
done
