#!/bin/bash
# round 6, visit w18: after removing the single-chunk specialisation / two-pass staging again (order pin kept): encoder + model tests, frame time
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
out=$R/gpurun_out/lean_final_r6w18.txt; : > $out
timeout 600 python -m pytest tests/test_hip_encoder.py tests/test_model_boundary.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3 | tee -a $out
for m in m s sc; do
  timeout 200 python bench.py --model $m --steps 300 --warmup 30 --no-cpu-baseline --no-extras 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$m', d['value'], d['ms_per_step'])" | tee -a $out
done
