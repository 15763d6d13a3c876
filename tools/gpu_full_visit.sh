#!/bin/bash
# The full measurement visit (rounds 2-3): [tests] -> PMC traffic passes of the bench command -> bench.py (default model, with the
# measured traffic) -> bench.py for the other BASELINE configs -> rocprofv3 kernel stats of the bench command.
#   gpurun --timeout 1500 -- 'bash tools/gpu_full_visit.sh <tag> [tests|notests] [full|quick]'
tag=${1:-x}; tests=${2:-tests}; mode=${3:-full}
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "$tests" = tests ]; then
  timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu_$tag.log 2>&1
  echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$tag.log
  grep -E "passed|failed" gpurun_out/pytest_gpu_$tag.log | tail -2
  grep -E "^(FAILED|ERROR)" gpurun_out/pytest_gpu_$tag.log | head -20
fi
# HBM traffic of the benched decoder kernels: separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass)
rm -rf /tmp/pmc_bench; mkdir -p /tmp/pmc_bench
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_bench/$c -- python $R/bench.py --no-extras --steps 10 --warmup 3 --repeats 1 --no-graph > /tmp/pmc_bench_$c.log 2>&1 )
done
timeout 600 python bench.py --traffic-dir /tmp/pmc_bench > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
echo "bench rc=$?"; cut -c1-600 gpurun_out/bench_$tag.json
python - <<PY
import json
try:
    d = json.load(open('gpurun_out/bench_$tag.json'))
    print('value', d['value'], 'roofline', d.get('roofline'), 'parity', d.get('parity'), 'protocol', d.get('fps_reference_protocol'))
    print('decoder', d['decoder']['us_per_batch_eager'], [ (l['kernel'], l['avg_us']) for l in d['decoder']['launches'] if l['in_decoder']])
except Exception as e:
    print('bench parse failed', e); print(open('gpurun_out/bench_$tag.err').read()[-1500:])
PY
if [ "$mode" = full ]; then
  # HyperSeg-L: HBM traffic of its dominant launch as well
  rm -rf /tmp/pmc_bench_l; mkdir -p /tmp/pmc_bench_l
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_bench_l/$c -- python $R/bench.py --model l --no-extras --steps 4 --warmup 2 --repeats 1 --no-graph > /tmp/pmc_bench_l_$c.log 2>&1 )
  done
  for m in s l sc; do
    extra=""; [ "$m" = l ] && extra="--traffic-dir /tmp/pmc_bench_l"
    timeout 600 python bench.py --model $m --steps 20 --warmup 5 $extra > gpurun_out/bench_${tag}_$m.json 2> gpurun_out/bench_${tag}_$m.err
    python - <<PY
import json
try:
    d = json.load(open('gpurun_out/bench_${tag}_$m.json'))
    print('$m', d['value'], d['ms_per_step'], d.get('roofline'), d.get('parity'))
    print('   decoder', d['decoder']['us_per_batch_eager'], [ (l['kernel'][3:-4], l['avg_us']) for l in d['decoder']['launches'] if l['in_decoder']])
except Exception as e:
    print('$m bench parse failed', e); print(open('gpurun_out/bench_${tag}_$m.err').read()[-1500:])
PY
  done
  ( cd /tmp && rm -rf /tmp/prof_bench && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $R/bench.py --no-cpu-baseline > /tmp/prof_bench.log 2>&1
    f=$(find /tmp/prof_bench -name '*kernel_stats.csv' | head -1)
    if [ -n "$f" ]; then cp "$f" $R/gpurun_out/bench_kernel_stats_$tag.csv; python $R/tools/kstats.py "$f" "" 14; else echo "no stats"; tail -5 /tmp/prof_bench.log; fi )
fi
cd $R && python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
