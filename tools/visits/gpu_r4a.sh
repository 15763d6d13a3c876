#!/bin/bash
# Round-4 visit a: full GPU suite, the default bench line (self-spawned PMC traffic passes), one-rank collective probes.
tag=${1:-r4a}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu_$tag.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu_$tag.log; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_gpu_$tag.log | head
t0=$(date +%s)
timeout 900 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s"
python - <<PY
import json
try:
    d = json.load(open('gpurun_out/bench_$tag.json'))
    print('value', d['value'], d['ms_per_step'], d['dtype'])
    print('roofline', d.get('roofline'))
    for k in ('exact_f32', 'split_f16', 'library_gemm_f32', 'parity', 'cpu_baseline'):
        print(k, d.get(k))
    print('decoder', d['decoder']['us_per_batch_eager'], [(l['kernel'][3:-4], l['avg_us']) for l in d['decoder']['launches'] if l['in_decoder']])
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/bench_$tag.err').read()[-2500:])
PY
for probe in "ingraph 0" "ingraph 7" "allgather 0" "allgather 7" "direct 0"; do
  set -- $probe
  timeout 300 python bench.py --no-extras --steps 200 --warmup 20 --repeats 3 --collective $1 --probe-load $2 2> gpurun_out/probe_${tag}_$1_$2.err | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); c=d['collective']; print('probe $1 load $2:', d['ms_per_step'], 'without', c['ms_per_step_without'], 'overhead %', c['overhead_pct'], 'policy', c['policy'], c['notes'])
except Exception as e: print('probe $1 $2 failed', e)
" | tee -a gpurun_out/probe_$tag.txt
  tail -2 gpurun_out/probe_${tag}_$1_$2.err | cut -c1-300
done
