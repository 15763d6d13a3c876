#!/bin/bash
# round 6, visit 16: CamVid HyperSeg-L's two inverted-residual shapes in the exact-f32 fused kernel's table (they fell to the generic kernel: 451 us)
tag=${1:-r6v16}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "full_config and Lc" 2>&1 | tail -3
timeout 400 python bench.py --model lc --steps 100 --warmup 10 --no-cpu-baseline --traffic off > gpurun_out/bench_lc_$tag.json 2> gpurun_out/bench_lc_$tag.err
python -c "
import json; d=json.load(open('gpurun_out/bench_lc_$tag.json')); print('lc', d['value'], d['ms_per_step'], 'exact_f32', d.get('exact_f32'))" | cut -c1-600
