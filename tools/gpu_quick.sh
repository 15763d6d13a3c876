#!/bin/bash
# Short GPU visit: a subset of the GPU tests, then timing-only bench lines.
#   gpurun --timeout 400 -- 'bash tools/gpu_quick.sh <tag> "<pytest -k expr>" [models...]'
tag=${1:-x}; expr=${2:-se_gate}; shift 2; models=${@:-m}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider -k "$expr" 2>&1 | tail -4
for m in $models; do
  timeout 200 python bench.py --model $m --no-extras --no-cpu-baseline --steps 200 --warmup 20 --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$m', d['value'], d['ms_per_step'], d['repeats']['ms_per_step'])" | tee -a gpurun_out/quick_$tag.txt
done
