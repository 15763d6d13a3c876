#!/bin/bash
# round 5, last visit: the full GPU suite on the final code, smoke(), the default bench line once more
tag=${1:-r5fd}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu_$tag.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$tag.log
grep -E "passed|failed" gpurun_out/pytest_gpu_$tag.log | tail -2
grep -E "^(FAILED|ERROR)" gpurun_out/pytest_gpu_$tag.log | head -20 | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 700 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
echo "bench rc=$?"
python - <<PY
import json
d = json.load(open('gpurun_out/bench_$tag.json'))
print('value', d['value'], d['ms_per_step'], 'roofline', {k: d['roofline'].get(k) for k in ('frac', 'avg_launch_us', 'traffic')})
o = d.get('other_configs', {})
print('s', o.get('s', {}).get('value'), 'train', o.get('train_sc', {}).get('fp32'), o.get('train_sc', {}).get('bf16'))
PY
