#!/bin/bash
# round 6, visit w14: the split GEMM's two-chunk form for 1280 < K <= 2560 (HyperSeg-M's last project conv, K = 1920, was the frame's one
# library GEMM): tests, whole-frame A/B (HS_SPLIT_GEMM_MAX_K=1280 | 2560)
#   gpurun --timeout 1200 -- 'bash tools/visits/r6w14.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
out=$R/gpurun_out/split_k1920_r6w14.txt; : > $out
timeout 600 python -m pytest tests/test_split_gemm.py tests/test_hip_encoder.py tests/test_model_boundary.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -5 | tee -a $out
for round in 1 2 3; do
  for k in 1280 2560; do
    HS_SPLIT_GEMM_MAX_K=$k timeout 200 python bench.py --model m --steps 300 --warmup 30 --no-cpu-baseline --no-extras 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$round m max_k=$k', d['value'], d['ms_per_step'])" | tee -a $out
  done
done
