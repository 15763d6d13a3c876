#!/bin/bash
# Round-4 full measurement visit: whole GPU suite -> bench.py (default model; it measures its own HBM traffic) -> the other BASELINE
# configs -> rocprofv3 kernel stats of the bench command -> SQ counters of the decoder kernels -> smoke.
#   gpurun --timeout 1800 -- 'bash tools/gpu_r4_final.sh <tag>'
tag=${1:-r4z}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu_$tag.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/pytest_gpu_$tag.log | tail -2; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_gpu_$tag.log | head -20
t0=$(date +%s)
timeout 900 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s"
python - <<PY
import json
try:
    d = json.load(open('gpurun_out/bench_$tag.json'))
    print('value', d['value'], d['ms_per_step'], '|', d['dtype'])
    print('roofline', d.get('roofline'))
    for k in ('exact_f32', 'split_f16', 'library_gemm_f32', 'parity', 'cpu_baseline', 'fps_reference_protocol'):
        print(k, d.get(k))
    print('decoder', d['decoder']['us_per_batch_eager'], [(l['kernel'][3:-4], l['avg_us']) for l in d['decoder']['launches'] if l['in_decoder']])
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/bench_$tag.err').read()[-2500:])
PY
for m in s sc l; do
  timeout 600 python bench.py --model $m --steps 20 --warmup 5 --traffic off > gpurun_out/bench_${tag}_$m.json 2> gpurun_out/bench_${tag}_$m.err
  python - <<PY
import json
try:
    d = json.load(open('gpurun_out/bench_${tag}_$m.json'))
    print('$m', d['value'], d['ms_per_step'], d.get('roofline', {}).get('frac'), d.get('parity'))
    print('   decoder', d['decoder']['us_per_batch_eager'], [(l['kernel'][3:-4], l['avg_us']) for l in d['decoder']['launches'] if l['in_decoder']])
except Exception as e:
    print('$m bench parse failed', e); print(open('gpurun_out/bench_${tag}_$m.err').read()[-1500:])
PY
done
( cd /tmp && rm -rf /tmp/prof_bench && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $R/bench.py --no-cpu-baseline --traffic off > /tmp/prof_bench.log 2>&1
  f=$(find /tmp/prof_bench -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then cp "$f" $R/gpurun_out/bench_kernel_stats_$tag.csv; python $R/tools/kstats.py "$f" "" 14; else echo "no stats"; tail -5 /tmp/prof_bench.log; fi )
HS_IR_MATH=auto PMC_ITERS=20 bash tools/pmc_decoder.sh $tag > /dev/null 2>&1
for i in 1 2 3; do echo "== pmc pass $i"; grep -A9 "patch_irc_kernel" gpurun_out/pmc_${tag}_$i.txt | head -10; done
timeout 200 python tools/train_step_time.py 20 2>&1 | grep -v Warn | tail -4 | tee gpurun_out/train_step_$tag.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
