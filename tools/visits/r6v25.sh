#!/bin/bash
# round 6, visit 25: GraphedTrainStep owns the root gradient (one launch less per replay): the graphed-step tests and the config-5 step time
tag=${1:-r6v25}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_training.py -m gpu -q -p no:cacheprovider -x -k "graphed or config5 or optimizer or validation" 2>&1 | tail -3
timeout 200 python tools/train_step_time.py 40 graph graph_bf16 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/train_step_$tag.txt
