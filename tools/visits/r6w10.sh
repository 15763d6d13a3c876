#!/bin/bash
# round 6, visit w10: the early blocks' squeeze-excite gate as one single-workgroup launch (se_gate_early_kernel): tests, whole-frame A/B
#   gpurun --timeout 1200 -- 'bash tools/visits/r6w10.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
out=$R/gpurun_out/se_early_r6w10.txt; : > $out
timeout 600 python -m pytest tests/test_hip_encoder.py tests/test_model_boundary.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -5 | tee -a $out
for round in 1 2; do
  for m in m; do
    for e in 0 1; do
      HS_SE_EARLY=$e timeout 200 python bench.py --model $m --steps 300 --warmup 30 --no-cpu-baseline --no-extras 2>/dev/null | \
        python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$round $m se_early=$e', d['value'], d['ms_per_step'])" | tee -a $out
    done
  done
done
