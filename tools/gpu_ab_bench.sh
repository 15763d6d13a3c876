#!/bin/bash
# Same-box A/B of whole-frame bench.py between the product library and variant libraries (tools/build_variants.py), interleaved
# twice so that box-to-box and clock drift do not decide the comparison.
#   gpurun --timeout 500 -- '[BENCH_ARGS="--model s"] bash tools/gpu_ab_bench.sh <tag> [variant names...]'
tag=${1:-x}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
out=$R/gpurun_out/ab_bench_$tag.txt; : > $out
for round in 1 2; do
  for v in product "$@"; do
    lib=$R/hyperseg_amd/lib/libhyperseg_hip_$v.so; [ $v = product ] && lib=$R/hyperseg_amd/lib/libhyperseg_hip.so
    [ -f $lib ] || { echo "missing $lib" | tee -a $out; continue; }
    HS_HIP_LIB=$lib timeout 120 python bench.py $BENCH_ARGS --steps ${BENCH_STEPS:-300} --warmup 30 --no-cpu-baseline --no-extras 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$round $v', d['value'], d['ms_per_step'])" | tee -a $out
  done
done
