#!/bin/bash
# round 6, visit w17: the lean kernel's single-chunk specialisation (ONE: 146 -> 89 registers at the 24 -> 144 blocks) + two-pass output staging
# (43 -> 33 KB of LDS: 4 workgroups per CU): tests, per block and whole frame against HS_MBX_ONE=0
#   gpurun --timeout 1200 -- 'bash tools/visits/r6w17.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp HS_BENCH_MBCONV_FIRST_TABLE_ONLY=1
out=$R/gpurun_out/lean_one_r6w17.txt; : > $out
timeout 600 python -m pytest tests/test_hip_encoder.py tests/test_model_boundary.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -5 | tee -a $out
run() { echo "== $*" | tee -a $out; env "$@" timeout 120 python tools/bench_mbconv.py 2>&1 | tail -12 | grep -E "^ +[2-8] |sum" | cut -c1-75 | tee -a $out; }
run HS_MBX_ONE=0
run HS_MBX_ONE=1
run HS_MBX_ONE=1 HS_MBX_MIN_WG=1024
run HS_MBX_ONE=1 HS_MBX_MIN_WG=1536
run HS_MBX_ONE=1 HS_MBX_OTH1=8 HS_MBX_MIN_WG=1536
for round in 1 2; do
  for m in m s; do
    for one in 0 1; do
      HS_MBX_ONE=$one timeout 200 python bench.py --model $m --steps 300 --warmup 30 --no-cpu-baseline --no-extras 2>/dev/null | \
        python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$round $m one=$one', d['value'], d['ms_per_step'])" | tee -a $out
    done
  done
done
