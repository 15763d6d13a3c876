#!/bin/bash
# round 6, visit 5: blocked signal2weights with 16-byte stores (aligned quads of the bank rows, block rows staged at the row's own
# alignment): the whole GPU suite (banks are bit-identical by construction), per-kernel decoder times M / S / Sc / L / Lc.
tag=${1:-r6v5}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu_$tag.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$tag.log
grep -E "passed|failed" gpurun_out/pytest_gpu_$tag.log | tail -2
grep -E "^(FAILED|ERROR)" gpurun_out/pytest_gpu_$tag.log | head -20 | cut -c1-300
export HS_IR_MATH=auto
for cfg in M S Sc L Lc; do
  rm -rf /tmp/pv; cd /tmp
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -- python $R/tools/decoder_loop.py $cfg 30 > /tmp/pv.log 2>&1
  cd $R; echo "== $cfg" | tee -a gpurun_out/decoder_kernels_$tag.txt; grep decoder /tmp/pv.log | tee -a gpurun_out/decoder_kernels_$tag.txt
  f=$(find /tmp/pv -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && python tools/kstats.py $f hs:: 12 | cut -c1-150 | tee -a gpurun_out/decoder_kernels_$tag.txt
done
