#!/bin/bash
# split-arithmetic IR kernel: parity (both modes) + decoder kernel stats in both modes
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
tag=${1:-r2p}
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -p no:cacheprovider -x -k "${2:-split or full_config or op_d or inverted or tiny}" 2>&1 | tail -15
export TMPDIR=/tmp
for math in split f32; do
for cfg in M L; do
  rm -rf /tmp/prof_$cfg
  ( cd /tmp && HS_IR_MATH=$math timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$cfg -- python $R/tools/decoder_loop.py $cfg 20 > /tmp/prof_$cfg.log 2>&1 )
  echo "== $math $cfg"; grep -E "decoder (eager|graph)" /tmp/prof_$cfg.log
  f=$(find /tmp/prof_$cfg -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/kstats_${tag}_${cfg}_$math.csv && python tools/kstats.py "$f" "hs::" 8
done
done
