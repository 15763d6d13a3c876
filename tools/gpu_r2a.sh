#!/bin/bash
# Round-2 GPU visit A: all GPU tests (no -x), then per-kernel decoder timings of the product library and the dev variants.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
tag=${1:-r2a}
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu_$tag.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$tag.log
grep -E "passed|failed|error" gpurun_out/pytest_gpu_$tag.log | tail -3
grep -E "^(FAILED|ERROR)" gpurun_out/pytest_gpu_$tag.log | head -30
export TMPDIR=/tmp
prof() {  # name lib cfg iters [batch]
  local name=$1 lib=$2; shift 2
  rm -rf /tmp/prof_$name
  ( cd /tmp && HS_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python $R/tools/decoder_loop.py "$@" > /tmp/prof_$name.log 2>&1 )
  grep -E "decoder (eager|graph)" /tmp/prof_$name.log
  f=$(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then cp "$f" gpurun_out/kstats_${tag}_$name.csv; python tools/kstats.py "$f" "" 12; else echo "no stats for $name"; tail -5 /tmp/prof_$name.log; fi
}
L=$R/hyperseg_amd/lib
echo "=== M product";  prof M_new   $L/libhyperseg_hip.so          M 50
echo "=== M oldir";    prof M_old   $L/libhyperseg_hip_oldir.so    M 50
echo "=== M nointer";  prof M_noint $L/libhyperseg_hip_nointer.so  M 50
echo "=== S product";  prof S_new   $L/libhyperseg_hip.so          S 50
echo "=== Sc product"; prof Sc_new  $L/libhyperseg_hip.so          Sc 50
echo "=== L product";  prof L_new   $L/libhyperseg_hip.so          L 10
