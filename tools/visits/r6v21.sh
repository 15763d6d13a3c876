#!/bin/bash
# round 6, visit 21: two GraphedModel wrappers replayed from two threads with the chain on (ChainGate in the graph path)
tag=${1:-r6v21}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_encoder.py -m gpu -q -p no:cacheprovider -x -k "two_graphed_models or graphed_model_serves" 2>&1 | tail -15 | cut -c1-300
