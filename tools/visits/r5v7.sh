#!/bin/bash
# round 5, visit 7: the squeeze-excite gate finished by the pooling launch's last workgroups (csrc/hs_se_tail.h): tests, frame A/B, kernel times
tag=${1:-r5v7}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_encoder.py -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_encoder_$tag.log 2>&1
echo "encoder pytest rc=$?"; tail -4 gpurun_out/pytest_encoder_$tag.log | cut -c1-300
grep -E "^E  " gpurun_out/pytest_encoder_$tag.log | head -30 | cut -c1-400
bash tools/gpu_ab_env.sh se_$tag "HS_SE_TAIL=1" "HS_SE_TAIL=0"
out=$R/gpurun_out/se_tail_kernels_$tag.txt; : > $out
for v in 1 0; do
  rm -rf /tmp/pv; cd /tmp
  HS_SE_TAIL=$v timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -- python $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > /tmp/pv.log 2>&1
  cd $R; echo "== HS_SE_TAIL=$v" | tee -a $out
  f=$(find /tmp/pv -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && python tools/kstats.py $f hs:: 40 | cut -c1-160 | tee -a $out
done
