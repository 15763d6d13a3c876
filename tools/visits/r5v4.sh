#!/bin/bash
# round 5, visit 4: thread test after the packed-image lifetime fix, whole GPU suite, chain kernel phase stamps + per-kernel times
tag=${1:-r5v4}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu_$tag.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$tag.log
grep -E "passed|failed" gpurun_out/pytest_gpu_$tag.log | tail -2
grep -E "^(FAILED|ERROR)" gpurun_out/pytest_gpu_$tag.log | head -20 | cut -c1-300
grep -E "^E  " gpurun_out/pytest_gpu_$tag.log | head -10 | cut -c1-400
out=$R/gpurun_out/chain_$tag.txt; : > $out
HS_K1_CHAIN=1 HS_HIP_LIB=$R/hyperseg_amd/lib/libhyperseg_hip_stamps_kc.so timeout 120 python tools/kc_phase_times.py M 2>&1 | tee -a $out
HS_K1_CHAIN=1 HS_HIP_LIB=$R/hyperseg_amd/lib/libhyperseg_hip_stamps_kc.so timeout 120 python tools/kc_phase_times.py Sc 2>&1 | tee -a $out
export HS_IR_MATH=auto
for cfg in M Sc; do
  rm -rf /tmp/pv; cd /tmp
  HS_K1_CHAIN=1 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -- python $R/tools/decoder_loop.py $cfg 60 > /tmp/pv.log 2>&1
  cd $R; echo "== chain ($cfg)" | tee -a $out; grep "decoder" /tmp/pv.log | tee -a $out
  f=$(find /tmp/pv -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && python tools/kstats.py $f hs:: 12 | cut -c1-150 | tee -a $out
done
