#!/bin/bash
# One GPU-box visit: parity tests, bench, per-dispatch trace of one graph-replayed frame.  Usage (from the repo root):
#   gpurun --timeout 900 -- 'bash tools/gpu_round.sh <tag> [tests|notests] [trace|notrace]'
tag=${1:-x}; tests=${2:-tests}; trace=${3:-trace}; mbx=${4:-nombx}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
if [ "$tests" = tests ]; then
  timeout 700 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu_$tag.log 2>&1
  echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$tag.log
  tail -5 gpurun_out/pytest_gpu_$tag.log
fi
if [ "$mbx" = mbx ]; then
  timeout 200 python tools/bench_mbconv.py > gpurun_out/mbconv_$tag.txt 2>&1
  tail -25 gpurun_out/mbconv_$tag.txt
fi
timeout 300 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
echo "bench rc=$?"; cat gpurun_out/bench_$tag.json | cut -c1-400
# extra bench variants: arguments 5.. are "name:ENV1=V1,ENV2=V2" (no spaces); each runs bench.py --no-cpu-baseline
shift 4 2>/dev/null || shift $#
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  ( IFS=,; for kv in $envs; do export "$kv"; done
    timeout 200 python bench.py --no-cpu-baseline > gpurun_out/bench_${tag}_$name.json 2> gpurun_out/bench_${tag}_$name.err )
  echo "variant $name ($envs): $(python -c "import json,sys; d=json.loads(open('gpurun_out/bench_${tag}_$name.json').read() or '{}'); print(d.get('value'), d.get('ms_per_step'))" 2>&1 | tail -1)"
done
if [ "$mbx" = final ]; then
  timeout 400 python tools/fps_configs.py > gpurun_out/fps_configs_$tag.jsonl 2> gpurun_out/fps_configs_$tag.err
  cat gpurun_out/fps_configs_$tag.jsonl
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_bench &&
    timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python $R/bench.py --no-cpu-baseline > /tmp/prof_bench.log 2>&1
    f=$(find /tmp/prof_bench -name '*kernel_stats.csv' | head -1)
    if [ -n "$f" ]; then cp "$f" $R/gpurun_out/bench_kernel_stats_$tag.csv; tail -1 /tmp/prof_bench.log | cut -c1-200 > $R/gpurun_out/bench_under_rocprof_$tag.json; else echo "no stats"; tail -5 /tmp/prof_bench.log; fi )
fi
if [ "$trace" = trace ]; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf /tmp/prof_seq
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_seq -- python $R/tools/prof_graph.py 10 dw > /tmp/prof_seq.log 2>&1
  f=$(find /tmp/prof_seq -name '*kernel_trace.csv' | head -1)
  if [ -n "$f" ]; then
    python $R/tools/frame_sequence.py "$f" 10 > $R/gpurun_out/frame_seq_$tag.txt 2>&1
    tail -3 $R/gpurun_out/frame_seq_$tag.txt
  else
    echo "no kernel trace"; tail -5 /tmp/prof_seq.log
  fi
fi
