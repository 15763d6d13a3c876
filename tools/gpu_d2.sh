#!/bin/bash
# Op D two-launch form (hs_patch_ir_d2.hip): parity, then the HyperSeg-L decoder's per-kernel times with it on / off.
#   gpurun --timeout 600 -- 'bash tools/gpu_d2.sh <tag>'
tag=${1:-x}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
[ -n "$SKIP_TESTS" ] || timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "op_d or config_l" > gpurun_out/d2_pytest_$tag.txt 2>&1; tail -5 gpurun_out/d2_pytest_$tag.txt
out=$R/gpurun_out/d2_kernels_$tag.txt; : > $out
for v in 1 0; do
  rm -rf /tmp/pv; cd /tmp
  HS_IR_D2=$v timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -- python $R/tools/decoder_loop.py L 10 > /tmp/pv.log 2>&1
  cd $R; echo "== HS_IR_D2=$v" | tee -a $out; tail -2 /tmp/pv.log | tee -a $out
  f=$(find /tmp/pv -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && python tools/kstats.py $f hs:: 400 | tee -a $out
done
