#!/bin/bash
# round 6, visit 11: whose turn on a CU (IrcArgs.prio 0 / 1 / 2 / 3 through the HS_IRC_PRIO dev knob), two interleaved repetitions per
# configuration: rocprofv3 averages of the f16-split inverted-residual launches
tag=${1:-r6v11}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp HS_IR_MATH=auto
out=$R/gpurun_out/irc_prio_$tag.txt; : > $out
for rep in 1 2; do
for cfg in M S Sc Lc; do
for mode in 0 1 2 3; do
  rm -rf /tmp/pv; cd /tmp
  HS_IRC_PRIO=$mode timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -- python $R/tools/decoder_loop.py $cfg 40 > /tmp/pv.log 2>&1
  cd $R; f=$(find /tmp/pv -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && echo "$cfg prio=$mode rep=$rep $(python tools/kstats.py $f patch_irc 4 | tr '\n' ' ' | cut -c1-200)" | tee -a $out
done; done; done
