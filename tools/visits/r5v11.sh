#!/bin/bash
# round 5, visit 11: signal2weights as a stream (several blocks per workgroup, next fill in flight): parity, kernel times M / S / L, frame A/B
tag=${1:-r5v11}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_parity.py tests/test_model_boundary.py -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_s2w_$tag.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/pytest_s2w_$tag.log | cut -c1-300
grep -E "^E  " gpurun_out/pytest_s2w_$tag.log | head -20 | cut -c1-400
out=$R/gpurun_out/s2w_stream_$tag.txt; : > $out
for lib in "" "$R/hyperseg_amd/lib/libhyperseg_hip_s2bstream0.so"; do
  for cfg in M S L; do
    rm -rf /tmp/pv; cd /tmp
    HS_HIP_LIB=$lib timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -- python $R/tools/decoder_loop.py $cfg 40 > /tmp/pv.log 2>&1
    cd $R; echo "== lib=${lib:-product} ($cfg)" | tee -a $out; grep "decoder" /tmp/pv.log | tee -a $out
    f=$(find /tmp/pv -name '*kernel_stats.csv' | head -1)
    [ -n "$f" ] && python tools/kstats.py $f signal2weights 30 | cut -c1-150 | tee -a $out
  done
done
bash tools/gpu_ab_env.sh s2bs_$tag "HS_S2B=stream" "HS_HIP_LIB=$R/hyperseg_amd/lib/libhyperseg_hip_s2bstream0.so"
