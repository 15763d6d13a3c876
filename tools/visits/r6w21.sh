#!/bin/bash
# round 6, visit w21: block 0's project conv + block 1's depthwise half as one launch (hs_project_dw_fwd, PROJ form of the lean kernel): encoder /
# model tests, whole-frame A/B (HS_PROJECT_DW=0|1), interleaved
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
out=$R/gpurun_out/project_dw_r6w21.txt; : > $out
timeout 600 python -m pytest tests/test_hip_encoder.py tests/test_model_boundary.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -5 | tee -a $out
for round in 1 2 3; do
  for m in m s; do
    for e in 0 1; do
      HS_PROJECT_DW=$e timeout 200 python bench.py --model $m --steps 300 --warmup 30 --no-cpu-baseline --no-extras 2>/dev/null | \
        python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$round $m project_dw=$e', d['value'], d['ms_per_step'])" | tee -a $out
    done
  done
done
