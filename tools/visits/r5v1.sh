#!/bin/bash
# round 5, visit 1: the parity / advisor / boundary changes on the GPU -- whole GPU suite, the 200-seed sweep behind
# test_graphed_train_step_equals_eager, the default bench line with the new other_configs side objects.
tag=${1:-r5v1}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu_$tag.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$tag.log
grep -E "passed|failed" gpurun_out/pytest_gpu_$tag.log | tail -2
grep -E "^(FAILED|ERROR)" gpurun_out/pytest_gpu_$tag.log | head -20
timeout 300 python tools/graphed_step_seed_sweep.py 200 5 > gpurun_out/graphed_step_seed_sweep_$tag.txt 2> gpurun_out/graphed_step_seed_sweep_$tag.err
echo "sweep rc=$?"; tail -12 gpurun_out/graphed_step_seed_sweep_$tag.txt
timeout 600 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
echo "bench rc=$?"
python - <<PY
import json
try:
    d = json.load(open('gpurun_out/bench_$tag.json'))
    print('value', d['value'], d['ms_per_step'], 'roofline', {k: d['roofline'].get(k) for k in ('frac', 'avg_launch_us', 'traffic')})
    print('decoder', d['decoder']['us_per_batch_eager'], [(l['kernel'][3:-4], l['avg_us']) for l in d['decoder']['launches'] if l['in_decoder']])
    print('other_configs', json.dumps(d.get('other_configs'))[:3000])
    print('cpu', d.get('cpu_baseline'))
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/bench_$tag.err').read()[-2500:])
PY
