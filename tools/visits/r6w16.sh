#!/bin/bash
# round 6, visit w16: the lean kernel with a ragged last tile row (CamVid maps): encoder / model tests, CamVid-S / CamVid-L / HyperSeg-M frame A/B
# against HS_MBX_LEAN=0 for the ragged shapes' sake
#   gpurun --timeout 1200 -- 'bash tools/visits/r6w16.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
out=$R/gpurun_out/lean_ragged_r6w16.txt; : > $out
timeout 600 python -m pytest tests/test_hip_encoder.py tests/test_model_boundary.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -5 | tee -a $out
for round in 1 2; do
  for m in sc m; do
    for lean in 0 1; do
      HS_MBX_LEAN=$lean timeout 200 python bench.py --model $m --steps 300 --warmup 30 --no-cpu-baseline --no-extras 2>/dev/null | \
        python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$round $m lean=$lean', d['value'], d['ms_per_step'])" | tee -a $out
    done
  done
done
