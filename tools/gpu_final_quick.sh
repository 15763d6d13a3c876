#!/bin/bash
# Last short GPU visit of a round: the whole GPU test suite, then the bench line of the default model without the CPU leg.
#   gpurun --timeout 240 -- 'bash tools/gpu_final_quick.sh <tag>'
tag=${1:-x}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 170 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu_$tag.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$tag.log
tail -3 gpurun_out/pytest_gpu_$tag.log
timeout 80 python bench.py --no-cpu-baseline > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d = json.load(open('gpurun_out/bench_$tag.json'))
    print('value', d['value'], d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline']['avg_launch_us'], 'parity', d['parity'], 'protocol', d['fps_reference_protocol'])
except Exception as e:
    print('parse failed', e); print(open('gpurun_out/bench_$tag.err').read()[-800:])
PY
