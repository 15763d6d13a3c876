#!/bin/bash
# round 6, visit 2: the ADVICE r5 fixes on the GPU (chain gate + per-stream workspaces + error polling, Identity norms in train mode,
# Adam checkpoint round trip, the unify decoder's chain hook now live) and HyperSeg-S with the chain really on.
tag=${1:-r6v2}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "two_python_threads or k1_chain or identity_norms or adam or full_config or graphed or unify or serving or GraphedModel" > gpurun_out/pytest_$tag.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/pytest_$tag.log | cut -c1-300
grep -E "^(FAILED|ERROR)" gpurun_out/pytest_$tag.log | head -20 | cut -c1-300
for c in --chain-k1 --no-chain-k1 --chain-k1 --no-chain-k1; do
  timeout 200 python bench.py --model s --no-extras --steps 200 --warmup 30 --repeats 3 $c > /tmp/b.json 2>/tmp/b.err || tail -5 /tmp/b.err
  python -c "
import json; d=json.load(open('/tmp/b.json')); print('bench s $c', d['value'], d['ms_per_step'])" 2>&1 | tee -a gpurun_out/s_chain_ab_$tag.txt
done
