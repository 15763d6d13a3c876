#!/bin/bash
# round 6, visit x32: the whole GPU suite on the round's final tree (incl. the bf16 BatchNorm tests added after evidence visit r6gh) + smoke()
tag=${1:-r6x32}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/pytest_gpu_$tag.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$tag.log
tail -3 gpurun_out/pytest_gpu_$tag.log
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2
