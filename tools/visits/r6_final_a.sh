#!/bin/bash
# round 6, evidence visit A: whole GPU suite; default bench line (roofline + cpu_baseline + other_configs + legs_s); rocprofv3 kernel stats of
# the bench command; SQ counters of the decoder kernels (M, S); the other models' lines (s, sc, l, lc); smoke().
tag=${1:-r6fa}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu_$tag.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$tag.log
grep -E "passed|failed" gpurun_out/pytest_gpu_$tag.log | tail -2
grep -E "^(FAILED|ERROR)" gpurun_out/pytest_gpu_$tag.log | head -20 | cut -c1-300
( time timeout 700 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err ) 2>&1 | grep real
python - <<PY
import json
try:
    d = json.load(open('gpurun_out/bench_$tag.json'))
    print('value', d['value'], d['ms_per_step'], 'roofline', {k: d['roofline'].get(k) for k in ('frac', 'avg_launch_us', 'traffic', 'kernel')})
    print('decoder', d['decoder']['us_per_batch_eager'], [(l['kernel'][3:-4], l['avg_us']) for l in d['decoder']['launches'] if l['in_decoder']])
    print('legs', d['legs_s']); print('cpu', d['cpu_baseline'])
    o = d.get('other_configs', {})
    print('s', o.get('s', {}).get('value'), o.get('s', {}).get('decoder', {}).get('launches'), o.get('s', {}).get('roofline', {}).get('frac'))
    print('train', {k: o.get('train_sc', {}).get(k) for k in ('fp32', 'bf16', 'bf16_speedup_over_fp32', 'error')})
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/bench_$tag.err').read()[-2500:])
PY
( cd /tmp && rm -rf /tmp/prof_bench && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $R/bench.py --no-cpu-baseline --no-other-configs --traffic off > /tmp/prof_bench.log 2>&1
  f=$(find /tmp/prof_bench -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then cp "$f" $R/gpurun_out/bench_kernel_stats_$tag.csv; python $R/tools/kstats.py "$f" "" 16 | cut -c1-160; else echo "no stats"; tail -5 /tmp/prof_bench.log; fi )
for cfg in M S; do
  HS_K1_CHAIN=1 HS_IR_MATH=auto PMC_CFG=$cfg PMC_ITERS=12 bash tools/pmc_decoder.sh ${tag}_$cfg > /dev/null 2>&1
  echo "== PMC $cfg"; grep -A9 "patch_irc_kernel" gpurun_out/pmc_${tag}_${cfg}_1.txt | head -12
done
for m in s sc l lc; do
  timeout 400 python bench.py --model $m --steps 100 --warmup 10 --no-cpu-baseline --traffic off > gpurun_out/bench_${tag}_$m.json 2> gpurun_out/bench_${tag}_$m.err
  python - <<PY
import json
try:
    d = json.load(open('gpurun_out/bench_${tag}_$m.json'))
    print('$m', d['value'], d['ms_per_step'], {k: d['roofline'].get(k) for k in ('frac', 'avg_launch_us', 'kernel')}, d.get('parity'))
    print('   decoder', d['decoder']['us_per_batch_eager'], [(l['kernel'][3:-4], l['avg_us']) for l in d['decoder']['launches']])
except Exception as e:
    print('$m bench parse failed', e); print(open('gpurun_out/bench_${tag}_$m.err').read()[-1500:])
PY
done
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
