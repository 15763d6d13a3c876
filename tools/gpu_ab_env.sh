#!/bin/bash
# Same-box A/B of whole-frame bench.py between environment settings, interleaved twice.
#   gpurun --timeout 500 -- 'bash tools/gpu_ab_env.sh <tag> "HS_CTX_DOWN_SPLIT=1" "HS_CTX_DOWN_SPLIT=0"'
tag=${1:-x}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
out=$R/gpurun_out/ab_env_$tag.txt; : > $out
for round in 1 2; do
  for v in "$@"; do
    env $v timeout 120 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$round $v', d['value'], d['ms_per_step'])" | tee -a $out
  done
done
