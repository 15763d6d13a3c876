#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
tag=${1:-r2j}
timeout 900 python -m pytest tests/test_hip_training.py -m gpu -q -p no:cacheprovider > gpurun_out/pytest_train_$tag.log 2>&1
grep -E "passed|failed" gpurun_out/pytest_train_$tag.log | tail -2
grep -E "^(FAILED|ERROR)|above tolerance|Error" gpurun_out/pytest_train_$tag.log | cut -c1-1800 | head -20

