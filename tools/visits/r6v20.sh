#!/bin/bash
# round 6, visit 20: the N > 1 code path on one rank (collective probes: three ring slots = three captured graphs with the chain on)
tag=${1:-r6v20}; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
for pol in allgather ingraph direct gather; do
  timeout 300 python bench.py --collective $pol --no-extras --steps 100 --warmup 10 --repeats 3 > /tmp/b.json 2>/tmp/b.err || { echo "$pol FAILED"; tail -5 /tmp/b.err; }
  python -c "
import json; d=json.load(open('/tmp/b.json')); c=d['collective']; print('$pol', d['value'], d['ms_per_step'], 'overhead_pct', c['overhead_pct'], 'zero_copy', c['zero_copy'], 'completed', c['completed'])" | tee -a gpurun_out/collective_probes_$tag.txt
done
timeout 300 python bench.py --collective allgather --gather masks --no-extras --steps 100 --warmup 10 --repeats 3 > /tmp/b.json 2>/tmp/b.err || tail -5 /tmp/b.err
python -c "
import json; d=json.load(open('/tmp/b.json')); c=d['collective']; print('allgather masks', d['value'], d['ms_per_step'], c['overhead_pct'])" | tee -a gpurun_out/collective_probes_$tag.txt
python - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from hyperseg_amd import configs
from hyperseg_amd.utils.synthetic import fill_by_name
from hyperseg_amd.utils.inference import prepare_for_inference
m = fill_by_name(configs.build('hyperseg-m').eval(), seed=0)
prepare_for_inference(m, fold_bn=False, fused_depthwise=True, split_gemm=True)
m = m.cuda()
x = torch.rand(1, 3, 512, 1024, device='cuda')
with torch.no_grad():
    y0 = m(x).clone()
    graphs = []
    for k in range(3):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            y = m(x)
        graphs.append((g, y))
    for g, y in graphs * 3:
        g.replay()
    torch.cuda.synchronize()
    kc = m.decoder._k1_chain
    print('three captures back to back: equal to eager', all(torch.equal(y, y0) for _, y in graphs), 'captured_zero_fills', kc._pool.captured_zero_fills, 'error word', kc.error_word(), 'workspaces', len(kc._taken))
PY
