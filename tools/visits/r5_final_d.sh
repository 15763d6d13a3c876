#!/bin/bash
# round 5, last profile: rocprofv3 --kernel-trace --stats of the bench command on the final code
tag=${1:-r5fj}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/prof_bench && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $R/bench.py --no-cpu-baseline --no-other-configs --traffic off > /tmp/prof_bench.log 2>&1
  f=$(find /tmp/prof_bench -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then cp "$f" $R/gpurun_out/bench_kernel_stats_$tag.csv; python $R/tools/kstats.py "$f" "" 14 | cut -c1-150; else echo "no stats"; tail -5 /tmp/prof_bench.log; fi )
