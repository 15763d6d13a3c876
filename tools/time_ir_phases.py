"""Dev tool.  CPU: python tools/time_ir_phases.py build    GPU: python tools/time_ir_phases.py run
Per-phase s_memtime breakdown of the level-4 MFMA kernel (wave 0 of every workgroup)."""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
OUT = os.path.join(REPO, 'hyperseg_amd', 'lib', 'ablate')
LIB = os.path.join(OUT, 'libtiming.so')
if sys.argv[1] == 'build':
    from hyperseg_amd import build as B
    os.makedirs(OUT, exist_ok=True)
    objs = []
    for s in B.SOURCES:
        o = os.path.join(OUT, f'{s}.timing.o')
        subprocess.check_call([B._hipcc(), *B.FLAGS, '-DHS_IRM_TIMING=1', '-c', os.path.join(B.CSRC, s), '-o', o])
        objs.append(o)
    subprocess.check_call([B._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', *objs, '-o', LIB])
    print('built', LIB)
else:
    os.environ['HS_HIP_LIB'] = LIB
    sys.path.insert(0, os.path.join(REPO, 'tools'))
    import ctypes, torch, numpy as np
    from _workload import decoder_workload
    import hyperseg_amd._hip as hip
    d, x, s = decoder_workload('M')
    for _ in range(5):
        d(x, s)
    torch.cuda.synchronize()
    n = 512 * 32
    buf = (ctypes.c_longlong * n)()
    hip.lib.hs_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert hip.lib.hs_debug_read_stamps(buf, n) == 0
    st = np.array(buf, dtype=np.int64).reshape(512, 32)      # the LAST IR launch (level 4) overwrote level 3's stamps
    names = ['start->loads issued', 'prologue gathers+T', 'bank->LDS+barrier', 'Bfrag+barrier']
    d0 = st[:, 1:4] - st[:, 0:3]
    print('blocks: first start %d, last end %d cycles span (100 MHz ticks?)' % (0, int(st[:, 23].max() - st[:, 0].min())))
    print('kernel span per block (ticks): median %.0f' % np.median(st[:, 23] - st[:, 0]))
    for i, nm in enumerate(['bank loads issued', 'prologue gathers -> T in LDS', 'bank -> LDS + barrier', 'B frags + barrier']):
        if i == 0:
            print('%-32s %8.0f' % (nm, np.median(st[:, 1] - st[:, 0])))
    print('%-32s %8.0f' % ('all loads issued (0->24)', np.median(st[:, 24] - st[:, 0])))
    print('%-32s %8.0f' % ('wait + LDS stores (24->25)', np.median(st[:, 25] - st[:, 24])))
    print('%-32s %8.0f' % ('barrier (25->26)', np.median(st[:, 26] - st[:, 25])))
    print('%-32s %8.0f' % ('position passes (26->27)', np.median(st[:, 27] - st[:, 26])))
    print('%-32s %8.0f' % ('barrier (27->2)', np.median(st[:, 2] - st[:, 27])))
    print('%-32s %8.0f' % ('B frags + barrier (2->3)', np.median(st[:, 3] - st[:, 2])))
    for ch in range(5):
        b = 4 + 4 * ch
        nxt = st[:, b + 4] if ch < 4 else st[:, 22]
        print('chunk %d: operands+pw1 %6.0f  barrier %6.0f  dw %6.0f  barrier+?  pw3 %6.0f' % (
            ch, np.median(st[:, b + 1] - st[:, b]), np.median(st[:, b + 2] - st[:, b + 1]),
            np.median(st[:, b + 3] - st[:, b + 2]), np.median(nxt - st[:, b + 3])))
    print('%-32s %8.0f' % ('epilogue (22->23)', np.median(st[:, 23] - st[:, 22])))
    print('chunk0 start relative to kernel start', np.median(st[:, 4] - st[:, 0]))
