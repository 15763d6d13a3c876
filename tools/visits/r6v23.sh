#!/bin/bash
# round 6, visit 23: runtime (ROCclr) environment knobs against the graph-edge floor: the cost of a dependent node in a replayed graph
# (tools/graph_floor.py) and the whole frame (bench.py --no-extras), one setting per run
tag=${1:-r6v23}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
out=$R/gpurun_out/runtime_knobs_$tag.txt; : > $out
for v in "X=0" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "ROC_SKIP_KERNEL_ARG_COPY=1" "DEBUG_HIP_GRAPH_BATCH_SIZE=1024" "AMD_DIRECT_DISPATCH=0" "ROC_ACTIVE_WAIT_TIMEOUT=1000" "GPU_MAX_HW_QUEUES=1" "ROC_USE_FGS_KERNARG=0" "DEBUG_HIP_KERNARG_COPY_OPT=0" "X=0"; do
  echo "== $v" | tee -a $out
  env $v timeout 100 python tools/graph_floor.py 200 2>&1 | grep elements | head -2 | tee -a $out
  env $v timeout 150 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('   bench', d['value'], d['ms_per_step'])" | tee -a $out
done
