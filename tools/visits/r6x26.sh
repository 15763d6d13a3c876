#!/bin/bash
# round 6, visit x26: stress of the loss' in-launch tail (eager + replayed, bit-equal to the two-Function route) and the whole GPU suite twice more
tag=${1:-r6x26}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/stress_loss_tail.py 1500 2>&1 | tail -5 | tee gpurun_out/stress_loss_tail_$tag.txt
for i in 1 2; do timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -2 | tee -a gpurun_out/pytest_twice_$tag.txt; done
