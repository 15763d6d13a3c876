#!/bin/bash
# round 6, visit 7: bench lines of the other models on the current code (event-timed launch tables): l, s, sc, lc
tag=${1:-r6v7}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for m in l s sc lc; do
  timeout 400 python bench.py --model $m --no-cpu-baseline --traffic off > gpurun_out/bench_${m}_$tag.json 2> gpurun_out/bench_${m}_$tag.err || tail -5 gpurun_out/bench_${m}_$tag.err
  python -c "
import json; d=json.load(open('gpurun_out/bench_${m}_$tag.json')); print('$m', d['value'], d['ms_per_step'], d['roofline']['kernel'][:40], d['roofline']['avg_launch_us'], d['roofline']['frac'], 'flips', d['parity'].get('argmax_flips')); print('   ', [(l['kernel'][3:-4], l['avg_us']) for l in d['decoder']['launches'] if l['in_decoder']], d['decoder']['us_per_batch_eager'])"
done
