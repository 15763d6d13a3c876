#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
tag=${1:-r2f}
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu_$tag.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_gpu_$tag.log | tail -2
grep -E "^(FAILED|ERROR)|AssertionError|Error:" gpurun_out/pytest_gpu_$tag.log | head -20
export TMPDIR=/tmp
for spec in "on:64" "off:0"; do
  name=${spec%%:*}; v=${spec#*:}
  for cfg in M S; do
    rm -rf /tmp/prof_$cfg$name
    ( cd /tmp && HS_BANK_IN_CONSUMER_MAX_PIXELS=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$cfg$name -- python $R/tools/decoder_loop.py $cfg 30 > /tmp/prof_$cfg$name.log 2>&1 )
    echo "== $cfg bank-in-consumer $name"; grep -E "decoder (graph)" /tmp/prof_$cfg$name.log
    f=$(find /tmp/prof_$cfg$name -name '*kernel_stats.csv' | head -1)
    [ -n "$f" ] && cp "$f" gpurun_out/kstats_${tag}_${cfg}_$name.csv && python tools/kstats.py "$f" "hs::" 40
  done
done
