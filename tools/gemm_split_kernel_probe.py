"""Dev probe of tools/dev/hs_gemm_split.hip (the encoder's 1x1 convolutions as our own GEMM on the f16
matrix cores with split operands).  Not part of the product library.

    python tools/gemm_split_kernel_probe.py --build        # hipcc -> hyperseg_amd/lib/libhs_dev_gemm.so (no GPU needed)
    python tools/gemm_split_kernel_probe.py --emulate      # CPU emulation of the kernel's arithmetic vs float64 (no GPU needed)
    python tools/gemm_split_kernel_probe.py [--only 0,3,9] # on an MI355X: accuracy vs float64 and time vs torch.mm per shape

Shapes: the lean-route GEMMs of HyperSeg-M's prepared encoder at 1024x512 (M = Cout, K = Cin, N = pixels)."""
import ctypes as C
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'hyperseg_amd', 'lib', 'libhs_dev_gemm.so')
SRC = os.path.join(ROOT, 'tools', 'dev', 'hs_gemm_split.hip')
SHAPES = [(40, 240, 8192), (80, 240, 2048), (480, 80, 2048), (80, 480, 2048), (112, 480, 2048), (672, 112, 2048),
          (112, 672, 2048), (192, 672, 512), (1152, 192, 512), (192, 1152, 512), (320, 1152, 512), (1280, 320, 512)]


def build():
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared', SRC, '-o', LIB]
    subprocess.run(cmd, check=True)
    return LIB


def row_scale(m):
    """2^(141 - eb) with eb = clamp(biased exponent of m, 27, 254): m * scale < 2^15 (exp_of / scale_of of the kernels)."""
    eb = (m.contiguous().view(torch.int32) >> 23).clamp(27, 254)
    return torch.ldexp(torch.ones_like(m), 141 - eb), torch.ldexp(torch.ones_like(m), eb - 141)


def presplit(w, kp):
    """f32 (M, K) -> f16 pieces (M, Kp) of the row-scaled weight + the inverse row scales."""
    sc, inv = row_scale(w.abs().amax(1))
    ws = w * sc[:, None]
    hi = ws.half()
    lo = (ws - hi.float()).half()
    pad = kp - w.shape[1]
    if pad:
        hi, lo = torch.nn.functional.pad(hi, (0, pad)), torch.nn.functional.pad(lo, (0, pad))
    return hi.contiguous(), lo.contiguous(), inv.contiguous()


def plan(k):
    """(nwv, ks, Kp): waves per workgroup and 32-wide k-steps per wave with the least zero padding (ks <= 8)."""
    best = None
    for nwv in (8, 4, 2, 1):
        ks = -(-k // (32 * nwv))
        if ks <= 8:
            cand = (nwv * ks * 32 - k, -nwv, nwv, ks)
            best = cand if best is None or cand < best else best
    return best[2], best[3], best[2] * best[3] * 32


def plan_v1(k):
    """v1: (nwv, ks, Kp) with nwv in {2, 4, 8} waves, the fewest k-steps per wave (ks <= 5: K <= 1280)."""
    steps = -(-k // 32)
    nwv = 2
    while nwv < 8 and nwv < steps:
        nwv *= 2
    ks = -(-steps // nwv)
    return (nwv, ks, nwv * ks * 32) if ks <= 5 else None


def fragment_order(hi, lo, inv):
    """(M, Kp) f16 pieces -> [RT][KST][piece][lane = lrow + 16 kg][8] halfs (GemmArgsV1), rows padded to 16 RT."""
    m, kp = hi.shape
    rt = -(-m // 16)
    pad = 16 * rt - m
    if pad:
        hi, lo = torch.nn.functional.pad(hi, (0, 0, 0, pad)), torch.nn.functional.pad(lo, (0, 0, 0, pad))
        inv = torch.nn.functional.pad(inv, (0, pad), value=1.0)

    def sw(t):          # [RT][16 lrow][KST][4 kg][8 j] -> [RT][KST][kg][lrow][j]
        return t.view(rt, 16, kp // 32, 4, 8).permute(0, 2, 3, 1, 4)
    return torch.stack([sw(hi), sw(lo)], dim=2).contiguous(), inv.contiguous()


def emulate(w, x, gate):
    """The kernel's arithmetic on the CPU (f32 accumulation emulated in f64 of exactly representable f16 products)."""
    m, k = w.shape
    nwv, ks, kp = plan(k)
    hi, lo, inv = presplit(w, kp)
    xg = x * gate[:, None]
    xg = torch.nn.functional.pad(xg, (0, 0, 0, kp - k))
    y = torch.zeros(m, x.shape[1], dtype=torch.float64)
    for wv in range(nwv):
        sl = slice(wv * ks * 32, (wv + 1) * ks * 32)
        sc, invb = row_scale(xg[sl].abs().amax(0))
        xs = xg[sl] * sc[None, :]
        bh = xs.half()
        bl = (xs - bh.float()).half()
        ah, al = hi[:, sl].double(), lo[:, sl].double()
        part = al @ bh.double() + ah @ bl.double() + ah @ bh.double()
        y += (part.float() * invb[None, :]).double()
    return (y.float() * inv[:, None]).double()


def main():
    if '--build' in sys.argv:
        print(build())
        return
    if '--emulate' in sys.argv:
        g = torch.Generator().manual_seed(0)
        for m, k, n in SHAPES:
            n = min(n, 256)
            w = torch.randn(m, k, generator=g) / k ** 0.5
            x = torch.randn(k, n, generator=g) * torch.rand(k, 1, generator=g) * 4
            gate = torch.rand(k, generator=g)
            ref = w.double() @ (x.double() * gate.double()[:, None])
            e_split = float((emulate(w, x, gate) - ref).abs().max() / ref.abs().max())
            e_f32 = float(((w @ (x * gate[:, None])).double() - ref).abs().max() / ref.abs().max())
            print(f'M {m:5d} K {k:5d}: split emulation {e_split:.2e}   f32 matmul {e_f32:.2e}   plan {plan(k)}')
        return
    lib = C.CDLL(LIB)
    lib.hs_dev_gemm_split.restype = C.c_int
    lib.hs_dev_gemm_split.argtypes = [C.c_void_p] * 6 + [C.c_int32] * 8 + [C.c_void_p]
    lib.hs_dev_gemm_split_v1.restype = C.c_int
    lib.hs_dev_gemm_split_v1.argtypes = [C.c_void_p] * 5 + [C.c_int32] * 7 + [C.c_void_p]
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    only = [int(v) for v in sys.argv[sys.argv.index('--only') + 1].split(',')] if '--only' in sys.argv else range(len(SHAPES))
    for m, k, n in [SHAPES[i] for i in only]:
        nwv, ks, kp = plan(k)
        mt = 8 if m >= 128 else (4 if m >= 64 else 2)
        w = (torch.randn(m, k, generator=g) / k ** 0.5).to(dev)
        x = (torch.randn(k, n, generator=g) * torch.rand(k, 1, generator=g) * 4).to(dev)
        gate = torch.rand(1, k, generator=g).to(dev)
        hi, lo, inv = presplit(w, kp)
        y = torch.zeros(1, m, n, device=dev)

        def run(beta=0):
            st = lib.hs_dev_gemm_split(hi.data_ptr(), lo.data_ptr(), inv.data_ptr(), gate.data_ptr(), x.data_ptr(), y.data_ptr(),
                                       1, m, k, kp, n, beta, nwv, mt, torch.cuda.current_stream().cuda_stream)
            assert st == 0, st
        run()
        torch.cuda.synchronize()
        ref = w.double() @ (x.double() * gate[0].double()[:, None])
        e_split = float((y[0].double() - ref).abs().max() / ref.abs().max())
        wg = w * gate                                            # what se_excite hands to the library GEMM today
        y2 = torch.empty(m, n, device=dev)
        torch.mm(wg, x, out=y2)
        e_f32 = float((y2.double() - ref).abs().max() / ref.abs().max())
        y.zero_()
        run(); run(1)
        torch.cuda.synchronize()
        e_beta = float((y[0].double() - 2 * ref).abs().max() / ref.abs().max())

        def timed(fn, reps=20, inner=20):
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3):
                    fn()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(inner):
                    fn()
            for _ in range(3):
                gr.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                gr.replay()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / (reps * inner) * 1e6
        t_split = timed(run) if '--v1-only' not in sys.argv else float('nan')
        t_mm = timed(lambda: torch.mm(wg, x, out=y2))
        print(f'M {m:5d} K {k:5d} N {n:5d}  plan nwv {nwv} ks {ks} mt {mt}:  split {t_split:6.2f} us (err {e_split:.1e}, beta path {e_beta:.1e})'
              f'   torch.mm {t_mm:6.2f} us (err {e_f32:.1e})', flush=True)
        p1 = plan_v1(k)
        if p1 is None:
            continue
        nwv1, ks1, kp1 = p1
        hi1, lo1, inv1 = presplit(w, kp1)
        wsw, inv1 = fragment_order(hi1, lo1, inv1)
        y1 = torch.zeros(1, m, n, device=dev)

        def run1(beta=0):
            st = lib.hs_dev_gemm_split_v1(wsw.data_ptr(), inv1.data_ptr(), gate.data_ptr(), x.data_ptr(), y1.data_ptr(),
                                          1, m, k, kp1, n, beta, nwv1, torch.cuda.current_stream().cuda_stream)
            assert st == 0, st
        run1()
        torch.cuda.synchronize()
        e1 = float((y1[0].double() - ref).abs().max() / ref.abs().max())
        run1(1)
        torch.cuda.synchronize()
        e1b = float((y1[0].double() - 2 * ref).abs().max() / ref.abs().max())
        t1 = timed(run1)
        print(f'{"":24s}v1 plan nwv {nwv1} ks {ks1}:          split {t1:6.2f} us (err {e1:.1e}, beta path {e1b:.1e})', flush=True)


if __name__ == '__main__':
    main()
