#!/bin/bash
# round 5, visit 2: hs_k1_chain_fwd (levels 0-2 as one launch) -- parity first, under a short timeout of its own; then the whole GPU suite;
# per-kernel decoder times (rocprofv3) with the chain off / on and of the staggered level-4 variants; whole-frame A/B of the chain.
tag=${1:-r5v2}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -x -k "k1_chain or two_python_threads" > gpurun_out/pytest_chain_$tag.log 2>&1
echo "chain pytest rc=$?"; tail -5 gpurun_out/pytest_chain_$tag.log | cut -c1-300
grep -E "^(FAILED|ERROR)|Error|error_word|refused" gpurun_out/pytest_chain_$tag.log | head -20 | cut -c1-300
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu_$tag.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$tag.log
grep -E "passed|failed" gpurun_out/pytest_gpu_$tag.log | tail -2
grep -E "^(FAILED|ERROR)" gpurun_out/pytest_gpu_$tag.log | head -20 | cut -c1-300
out=$R/gpurun_out/variants_$tag.txt; : > $out
export HS_IR_MATH=auto
for v in product chain irc_stag_hi40 irc_stag_hi100 irc_stag_lo40 irc_stag_xcd40; do
  lib=$R/hyperseg_amd/lib/libhyperseg_hip_$v.so; chain=0
  [ $v = product ] && lib=$R/hyperseg_amd/lib/libhyperseg_hip.so
  [ $v = chain ] && { lib=$R/hyperseg_amd/lib/libhyperseg_hip.so; chain=1; }
  [ -f $lib ] || { echo "missing $lib" | tee -a $out; continue; }
  rm -rf /tmp/pv; cd /tmp
  HS_K1_CHAIN=$chain HS_HIP_LIB=$lib timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -- python $R/tools/decoder_loop.py M 60 > /tmp/pv.log 2>&1
  cd $R; echo "== $v (M)" | tee -a $out; grep "decoder" /tmp/pv.log | tee -a $out
  f=$(find /tmp/pv -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && python tools/kstats.py $f hs:: 12 | cut -c1-150 | tee -a $out
done
unset HS_IR_MATH
for rep in 1 2; do
  for c in --no-chain-k1 --chain-k1; do
    timeout 200 python bench.py --no-extras --steps 300 --warmup 30 --repeats 3 $c > /tmp/b.json 2>/tmp/b.err
    python -c "
import json; d=json.load(open('/tmp/b.json')); print('bench $c', d['value'], d['ms_per_step'], d['repeats']['ms_per_step'])" 2>&1 | tee -a $out
  done
done
