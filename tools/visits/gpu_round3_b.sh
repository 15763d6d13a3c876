#!/bin/bash
# Round-3 visit: whole GPU suite, bench line (default), the one-rank collective probes, per-kernel profile of the bench.
#   gpurun --timeout 600 -- 'bash tools/gpu_round3_b.sh <tag>'
tag=${1:-x}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 240 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu_$tag.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$tag.log
tail -3 gpurun_out/pytest_gpu_$tag.log
timeout 100 python bench.py --no-cpu-baseline > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err; echo "bench rc=$?"
for c in allgather direct; do
  timeout 100 python bench.py --no-cpu-baseline --no-extras --collective $c > gpurun_out/bench_${tag}_$c.json 2> gpurun_out/bench_${tag}_$c.err; echo "bench $c rc=$?"
done
python - <<PY
import json
for n in ('', '_allgather', '_direct'):
    try:
        d = json.load(open('gpurun_out/bench_$tag' + n + '.json'))
        print(n or 'default', 'value', d['value'], d['ms_per_step'], 'collective', d.get('collective'), 'roofline', {k: d['roofline'][k] for k in ('bound', 'frac', 'avg_launch_us', 'achieved')} if 'roofline' in d else None, 'parity', d.get('parity'))
    except Exception as e:
        print(n, 'parse failed', e); print(open('gpurun_out/bench_$tag' + n + '.err').read()[-600:])
PY
