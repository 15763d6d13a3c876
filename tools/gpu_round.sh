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
