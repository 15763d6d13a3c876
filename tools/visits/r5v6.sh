#!/bin/bash
# round 5, visit 6: the inverted residual inside the chain after its latency pass (deferred DMA, tap table, independent matrix-core chains)
tag=${1:-r5v6}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -k "k1_chain" > gpurun_out/pytest_chain_$tag.log 2>&1
echo "chain pytest rc=$?"; tail -4 gpurun_out/pytest_chain_$tag.log | cut -c1-300
grep -E "^E  " gpurun_out/pytest_chain_$tag.log | head -20 | cut -c1-400
out=$R/gpurun_out/chain_$tag.txt; : > $out
echo "== stamps, HS_K1_CHAIN_IR_DEFAULT=1" | tee -a $out
HS_K1_CHAIN_IR_DEFAULT=1 HS_K1_CHAIN=1 HS_HIP_LIB=$R/hyperseg_amd/lib/libhyperseg_hip_stamps_kc.so timeout 120 python tools/kc_phase_times.py M 2>&1 | grep -v amdgpu.ids | tee -a $out
export HS_IR_MATH=auto
for v in "1 0" "1 1"; do
  set -- $v
  for cfg in M Sc; do
    rm -rf /tmp/pv; cd /tmp
    HS_K1_CHAIN=$1 HS_K1_CHAIN_IR_DEFAULT=$2 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -- python $R/tools/decoder_loop.py $cfg 60 > /tmp/pv.log 2>&1
    cd $R; echo "== chain=$1 ir=$2 ($cfg)" | tee -a $out; grep "decoder" /tmp/pv.log | tee -a $out
    f=$(find /tmp/pv -name '*kernel_stats.csv' | head -1)
    [ -n "$f" ] && python tools/kstats.py $f hs:: 12 | cut -c1-150 | tee -a $out
  done
done
