#!/bin/bash
# round 6, visit 15: the tests added after evidence visit A (BankSlices two consumers, signal2weights stores against a sentinel), fps harness flags
tag=${1:-r6v15}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "two_consumers or stay_inside or bank_slices or signal2weights" 2>&1 | tail -4
timeout 300 python -m hyperseg_amd.fps --iterations 40 --prepare --trace 2>/dev/null | tail -1 | cut -c1-400
timeout 300 python -m hyperseg_amd.fps --iterations 40 --prepare --gpus 0 2>/dev/null | tail -1 | cut -c1-400
