#!/bin/bash
# round 5, evidence visit A: the whole GPU suite (module defaults), the whole suite again with the chain forced on for EVERY decoder call
# (HS_K1_CHAIN=1: the nn.Module mirror's default is off), the default bench line.
tag=${1:-r5fa}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu_$tag.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$tag.log
grep -E "passed|failed" gpurun_out/pytest_gpu_$tag.log | tail -2
grep -E "^(FAILED|ERROR)" gpurun_out/pytest_gpu_$tag.log | head -20 | cut -c1-300
grep -E "^E  " gpurun_out/pytest_gpu_$tag.log | head -10 | cut -c1-400
HS_K1_CHAIN=1 timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu_chain_on_$tag.log 2>&1
echo "pytest (HS_K1_CHAIN=1) rc=$?" >> gpurun_out/pytest_gpu_chain_on_$tag.log
grep -E "passed|failed" gpurun_out/pytest_gpu_chain_on_$tag.log | tail -2
grep -E "^(FAILED|ERROR)" gpurun_out/pytest_gpu_chain_on_$tag.log | head -20 | cut -c1-300
grep -E "^E  " gpurun_out/pytest_gpu_chain_on_$tag.log | head -10 | cut -c1-400
timeout 700 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
echo "bench rc=$?"
python - <<PY
import json
try:
    d = json.load(open('gpurun_out/bench_$tag.json'))
    print('value', d['value'], d['ms_per_step'], 'roofline', {k: d['roofline'].get(k) for k in ('frac', 'avg_launch_us', 'traffic')}, d['parity'])
    print('decoder', d['decoder']['us_per_batch_eager'], [(l['kernel'][3:-4], l['avg_us']) for l in d['decoder']['launches'] if l['in_decoder']])
    print('exact', d['exact_f32']['value'], 'lib', d['library_gemm_f32'].get('value'), 'protocol', d['fps_reference_protocol'])
    o = d.get('other_configs', {})
    print('s', o.get('s', {}).get('value'), o.get('s', {}).get('decoder', {}).get('launches'), o.get('s', {}).get('roofline', {}).get('frac'))
    print('train', {k: o.get('train_sc', {}).get(k) for k in ('fp32', 'bf16', 'bf16_speedup_over_fp32', 'error')})
    print('cpu', d.get('cpu_baseline'))
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/bench_$tag.err').read()[-2500:])
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
