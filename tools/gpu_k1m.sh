#!/bin/bash
# Batched tiny-patch Op A (hs_patch_conv_k1m.hip): parity, then the HyperSeg-L decoder's per-kernel times with it on / off.
tag=${1:-x}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "tiny_patches or config_l or full_config or meta_patch or hyper_patch" > gpurun_out/k1m_pytest_$tag.txt 2>&1; tail -5 gpurun_out/k1m_pytest_$tag.txt
out=$R/gpurun_out/k1m_kernels_$tag.txt; : > $out
for v in product k1m_off; do
  lib=$R/hyperseg_amd/lib/libhyperseg_hip_$v.so; [ $v = product ] && lib=$R/hyperseg_amd/lib/libhyperseg_hip.so
  rm -rf /tmp/pv; cd /tmp
  HS_HIP_LIB=$lib timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -- python $R/tools/decoder_loop.py L 10 > /tmp/pv.log 2>&1
  cd $R; echo "== $v" | tee -a $out; grep "decoder" /tmp/pv.log | tee -a $out
  f=$(find /tmp/pv -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && python tools/kstats.py $f hs:: 400 | head -12 | tee -a $out
done
