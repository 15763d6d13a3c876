#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
tag=${1:-r2i}
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu_$tag.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_gpu_$tag.log | tail -2
grep -E "^(FAILED|ERROR)|Error:" gpurun_out/pytest_gpu_$tag.log | head -20
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
python - <<PY
import json
d = json.load(open('gpurun_out/bench_$tag.json'))
print('value', d['value'], d['repeats'], 'parity', d.get('parity'))
PY
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/prof_seq && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_seq -- python $R/tools/prof_graph.py 10 dw > /tmp/prof_seq.log 2>&1
  f=$(find /tmp/prof_seq -name '*kernel_trace.csv' | head -1)
  if [ -n "$f" ]; then python $R/tools/frame_sequence.py "$f" 10 > $R/gpurun_out/frame_seq_$tag.txt 2>&1; tail -4 $R/gpurun_out/frame_seq_$tag.txt; else echo "no trace"; tail -5 /tmp/prof_seq.log; fi )
