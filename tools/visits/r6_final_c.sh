#!/bin/bash
# round 6, evidence visit on the round's LAST commit: whole GPU suite + the default bench line + smoke()
tag=${1:-r6fd}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu_$tag.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$tag.log
grep -E "passed|failed" gpurun_out/pytest_gpu_$tag.log | tail -2
grep -E "^(FAILED|ERROR)" gpurun_out/pytest_gpu_$tag.log | head -20 | cut -c1-300
( time timeout 700 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err ) 2>&1 | grep real
python - <<PY
import json
d = json.load(open('gpurun_out/bench_$tag.json'))
print('value', d['value'], d['ms_per_step'], 'roofline', {k: d['roofline'].get(k) for k in ('frac', 'avg_launch_us', 'traffic')})
o = d.get('other_configs', {})
print('s', o.get('s', {}).get('value'), 'train', {k: (o.get('train_sc', {}).get(k) or {}).get('ms_per_step') for k in ('fp32', 'bf16')}, o.get('train_sc', {}).get('launches'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
PY
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
