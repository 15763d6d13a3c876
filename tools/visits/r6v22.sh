#!/bin/bash
# round 6, visit 22: CamVid HyperSeg-L whole model against the reference-made fixture model_Lc.npz (encoder + context head + six-level decoder), segment()
tag=${1:-r6v22}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_model_boundary.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -12 | cut -c1-300
