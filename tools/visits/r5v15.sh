#!/bin/bash
# round 5, visit 15: output channels per thread of the stem convolution (4 / 8 / 16): kernel time and whole-frame A/B
tag=${1:-r5v15}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
out=$R/gpurun_out/stem_cg_$tag.txt; : > $out
for lib in "" "$R/hyperseg_amd/lib/libhyperseg_hip_stem_cg4.so" "$R/hyperseg_amd/lib/libhyperseg_hip_stem_cg16.so"; do
  rm -rf /tmp/pv; cd /tmp
  HS_HIP_LIB=$lib timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -- python $R/tools/prof_graph.py 40 dw > /tmp/pv.log 2>&1
  cd $R; echo "== lib=${lib:-product}" | tee -a $out
  f=$(find /tmp/pv -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && python tools/kstats.py $f stem 60 | cut -c1-150 | tee -a $out
done
HS_X=1 timeout 120 python -m pytest tests/test_hip_encoder.py -m gpu -q -p no:cacheprovider -k "stem" 2>&1 | tail -2
bash tools/gpu_ab_env.sh stem_$tag "HS_STEM=8" "HS_HIP_LIB=$R/hyperseg_amd/lib/libhyperseg_hip_stem_cg4.so" "HS_HIP_LIB=$R/hyperseg_amd/lib/libhyperseg_hip_stem_cg16.so"
