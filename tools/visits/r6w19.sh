#!/bin/bash
# round 6, visit w19: lean kernel instantiated for Cin = 32 (EfficientNet-B3 / HyperSeg-L): mbconv tests, HyperSeg-L bs 32 and CamVid-L frame A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
out=$R/gpurun_out/lean_b3_r6w19.txt; : > $out
timeout 300 python -m pytest tests/test_hip_encoder.py -m gpu -q -p no:cacheprovider -x -k "mbconv or stem" 2>&1 | tail -3 | tee -a $out
for round in 1 2; do
  for m in l lc; do
    for lean in 0 1; do
      HS_MBX_LEAN=$lean timeout 300 python bench.py --model $m --steps 60 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | \
        python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$round $m lean=$lean', d['value'], d['ms_per_step'])" | tee -a $out
    done
  done
done
