#!/bin/bash
# round 6, evidence visit B (the round's LAST code): whole GPU suite, default bench line, rocprofv3 kernel stats of the bench command for
# every model, smoke().
tag=${1:-r6fc}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu_$tag.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$tag.log
grep -E "passed|failed" gpurun_out/pytest_gpu_$tag.log | tail -2
grep -E "^(FAILED|ERROR)" gpurun_out/pytest_gpu_$tag.log | head -20 | cut -c1-300
( time timeout 700 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err ) 2>&1 | grep real
python - <<PY
import json
d = json.load(open('gpurun_out/bench_$tag.json'))
print('value', d['value'], d['ms_per_step'], 'roofline', {k: d['roofline'].get(k) for k in ('frac', 'avg_launch_us', 'traffic')})
print('legs', d['legs_s'])
o = d.get('other_configs', {})
print('s', o.get('s', {}).get('value'), 'train', {k: (o.get('train_sc', {}).get(k) or {}).get('ms_per_step') for k in ('fp32', 'bf16')})
PY
for m in m s sc l lc; do
  ( cd /tmp && rm -rf /tmp/prof_$m && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$m -- python $R/bench.py --model $m --no-cpu-baseline --no-other-configs --traffic off --steps 100 --warmup 10 > /tmp/prof_$m.json 2> /tmp/prof_$m.log
    f=$(find /tmp/prof_$m -name '*kernel_stats.csv' | head -1)
    if [ -n "$f" ]; then cp "$f" $R/gpurun_out/bench_kernel_stats_${m}_$tag.csv; echo "== $m"; python $R/tools/kstats.py "$f" "hs::" 60 | grep -E "patch_ir|signal2w|chain|upsample|patch_conv" | head -8 | cut -c1-150; else echo "no stats $m"; tail -3 /tmp/prof_$m.log; fi )
done
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
