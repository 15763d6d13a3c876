#!/bin/bash
# round 5, visit 20: dw_tiles_fwd_kernel<BN> with the statistics finalised after the tile requests: tests, kernel time, step time
tag=${1:-r5v20}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_training.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
( cd /tmp && rm -rf /tmp/pt && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -- python $R/tools/train_step_time.py 20 fp32 > /tmp/pt.log 2>&1
  f=$(find /tmp/pt -name '*kernel_stats.csv' | head -1); python $R/tools/kstats.py $f dw_tiles 200 | cut -c1-140 )
timeout 200 python tools/train_step_time.py 30 graph graph_bf16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/train_step_$tag.txt
