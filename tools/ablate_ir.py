"""Dev tool: build libhyperseg_hip variants with HS_IRM_ABLATE bits and time the level-4 kernel of each on the GPU box.
   CPU side: python tools/ablate_ir.py build      GPU side: python tools/ablate_ir.py run"""
import os, subprocess, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
VARIANTS = [0, 1, 2, 4, 8, 2 | 8, 1 | 2 | 4 | 8]
OUT = os.path.join(REPO, 'hyperseg_amd', 'lib', 'ablate')

if sys.argv[1] == 'build':
    from hyperseg_amd import build as B
    os.makedirs(OUT, exist_ok=True)
    for v in VARIANTS:
        objs = []
        for s in B.SOURCES:
            o = os.path.join(OUT, f'{s}.{v}.o')
            cmd = [B._hipcc(), *B.FLAGS, f'-DHS_IRM_ABLATE={v}', '-c', os.path.join(B.CSRC, s), '-o', o]
            if s != 'hs_patch_ir_mfma.hip' and os.path.exists(os.path.join(OUT, f'{s}.0.o')) and v != 0:
                o = os.path.join(OUT, f'{s}.0.o')
            else:
                subprocess.check_call(cmd)
            objs.append(o)
        subprocess.check_call([B._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', os.path.join(OUT, f'lib{v}.so')])
        print('built', v)
else:
    for v in VARIANTS:
        env = dict(os.environ, HS_HIP_LIB=os.path.join(OUT, f'lib{v}.so'))
        code = r'''
import sys, os, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tools"))
from _workload import decoder_workload
import hyperseg_amd.functional as HF
d, x, s = decoder_workload("M")
orig = HF.patch_ir; recs = []
def w(*a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = orig(*a, **k); e1.record(); recs.append((e0, e1)); return r
HF.patch_ir = w
for _ in range(40):
    torch.cuda._sleep(600000); d(x, s)
torch.cuda.synchronize()
ts = [a.elapsed_time(b) * 1e3 for a, b in recs[20:]]
l3 = sorted(ts[0::2]); l4 = sorted(ts[1::2])
print("ablate=%%2d  L3 %%6.1f us   L4 %%6.1f us" %% (%d, l3[len(l3)//2], l4[len(l4)//2]))
''' % (REPO, REPO, v)
        subprocess.run([sys.executable, '-c', code], env=env)
