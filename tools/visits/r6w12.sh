#!/bin/bash
# round 6, visit w12: pointwise_conv_kernel with pixel-interleaved tiles (16-byte loads / stores): encoder tests, whole-frame A/B (HS_PW_VEC=0|1)
#   gpurun --timeout 1200 -- 'bash tools/visits/r6w12.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
out=$R/gpurun_out/pw_vec_r6w12.txt; : > $out
timeout 600 python -m pytest tests/test_hip_encoder.py tests/test_model_boundary.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -5 | tee -a $out
for round in 1 2 3; do
  for m in m s; do
    for e in 0 1; do
      HS_PW_VEC=$e timeout 200 python bench.py --model $m --steps 300 --warmup 30 --no-cpu-baseline --no-extras 2>/dev/null | \
        python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$round $m pw_vec=$e', d['value'], d['ms_per_step'])" | tee -a $out
    done
  done
done
