#!/bin/bash
# round 6, visit w6: whole-frame A/B of the lean fused expand + depthwise kernel (HS_MBX_LEAN=0 | 1, interleaved twice, models m / s / sc),
# then the whole GPU suite on the product library
#   gpurun --timeout 1500 -- 'bash tools/visits/r6w6.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
out=$R/gpurun_out/frame_ab_lean_r6w6.txt; : > $out
for round in 1 2; do
  for m in m s sc; do
    for lean in 0 1; do
      HS_MBX_LEAN=$lean timeout 200 python bench.py --model $m --steps 300 --warmup 30 --no-cpu-baseline --no-extras 2>/dev/null | \
        python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$round $m lean=$lean', d['value'], d['ms_per_step'])" | tee -a $out
    done
  done
done
timeout 1000 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -5 | tee gpurun_out/pytest_gpu_r6w6.log
