#!/bin/bash
# round 6, visit 1: baseline on this round's box (whole GPU suite + default bench line) and the TIMING-ONLY level-4 variants
# (split phase removed / DMA in place / 16 x 8 regions at 4 workgroups per CU): what a producer-side split bank could buy.
tag=${1:-r6v1}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/pytest_gpu_$tag.log 2>&1
echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$tag.log
grep -E "passed|failed" gpurun_out/pytest_gpu_$tag.log | tail -2
grep -E "^(FAILED|ERROR)" gpurun_out/pytest_gpu_$tag.log | head -20 | cut -c1-300
bash tools/gpu_variants.sh $tag M irc_nosplit irc_inplace irc_rh8 irc_rh8_inplace > /dev/null 2>&1
cat gpurun_out/variants_$tag.txt | grep -E "==|patch_irc|decoder_chain|signal2w|upsample|patch_ir_fused" | cut -c1-160
( time timeout 600 python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err ) 2>&1 | grep real
python -c "
import json; d=json.load(open('gpurun_out/bench_$tag.json')); print(d['value'], d['ms_per_step'], d['roofline']); print({k:(v if not isinstance(v,dict) else '...') for k,v in d.get('other_configs',{}).items()})"
