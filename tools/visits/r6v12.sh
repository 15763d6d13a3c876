#!/bin/bash
# round 6, visit 12: HyperSeg-L bs 32, level 5 (hs_patch_ir_px): timing-only bound without its stores
tag=${1:-r6v12}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in product px_nostore product px_nostore; do
  lib=$R/hyperseg_amd/lib/libhyperseg_hip_$v.so; [ $v = product ] && lib=$R/hyperseg_amd/lib/libhyperseg_hip.so
  HS_HIP_LIB=$lib timeout 300 python bench.py --model l --no-cpu-baseline --traffic off --no-other-configs --steps 40 --warmup 5 --repeats 2 > /tmp/b.json 2>/tmp/b.err || tail -3 /tmp/b.err
  python -c "
import json; d=json.load(open('/tmp/b.json')); print('$v', d['value'], d['ms_per_step'], [(l['kernel'][3:-4], l['avg_us']) for l in d['decoder']['launches']])" | tee -a gpurun_out/px_nostore_$tag.txt
done
