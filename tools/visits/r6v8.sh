#!/bin/bash
# round 6, visit 8: non-temporal bank stores in the blocked signal2weights (L: 790 MB of banks per bs-32 batch; M: 31.8 MB that the
# consumers want to find in the caches): bench lines product vs s2b_nt, interleaved
tag=${1:-r6v8}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
for v in product s2b_nt; do
  lib=$R/hyperseg_amd/lib/libhyperseg_hip_$v.so; [ $v = product ] && lib=$R/hyperseg_amd/lib/libhyperseg_hip.so
  for m in l m; do
    HS_HIP_LIB=$lib timeout 300 python bench.py --model $m --no-cpu-baseline --traffic off --no-other-configs --steps 60 --warmup 10 --repeats 3 > /tmp/b.json 2>/tmp/b.err || tail -3 /tmp/b.err
    python -c "
import json; d=json.load(open('/tmp/b.json')); print('$v $m', d['value'], d['ms_per_step'], [(l['kernel'][3:-4], l['avg_us']) for l in d['decoder']['launches'] if 'signal2w' in l['kernel'] or 'patch_conv' in l['kernel'] or 'chain' in l['kernel']])" | tee -a gpurun_out/s2b_nt_ab_$tag.txt
  done
done
done
