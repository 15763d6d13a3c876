#!/bin/bash
# round 6, visit 24: flakiness check of the concurrency tests (chain gate, graph wrappers, thread tests): ten repetitions
tag=${1:-r6v24}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout 300 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "two_python_threads or two_graphed_models or k1_chain_equals" 2>&1 | tail -1
done
