"""Dev probe: library bf16 GEMMs on 3-piece split operands (6 products, f32 output) vs the f32 library GEMM, at the
prepared encoder's GEMM shapes.  python tools/gemm_split_probe.py"""
import time
import torch

dev = torch.device('cuda:0')


def pieces(x, n=3):
    out, r = [], x
    for _ in range(n):
        p = r.to(torch.bfloat16)
        out.append(p)
        r = r - p.float()
    return out


def bench(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e6


for (b, m, k, n) in [(32, 816, 136, 1024), (32, 136, 816, 1024), (32, 1392, 232, 256), (32, 232, 1392, 256), (32, 288, 48, 4096),
                     (32, 48, 288, 4096), (1, 1152, 192, 512), (1, 192, 1152, 512), (1, 240, 40, 8192)]:
    w = torch.randn(m, k, device=dev) * 0.1
    x = torch.randn(b, k, n, device=dev)
    wb = w.view(1, m, k).expand(b, m, k)
    ref = torch.bmm(wb.double(), x.double())
    y32 = torch.bmm(wb, x)
    t32 = bench(lambda: torch.bmm(wb, x))
    ah, am, al = pieces(w)
    a3 = torch.cat([ah, ah, ah], 1).view(1, m, 3 * k).expand(b, m, 3 * k)
    a2 = torch.cat([am, am], 1).view(1, m, 2 * k).expand(b, m, 2 * k)
    a1 = al.view(1, m, k).expand(b, m, k)
    bh, bm, bl = pieces(x)
    bp = torch.cat([bh, bm, bl], 1).contiguous()          # (b, 3k, n)

    def split_gemm():
        y = torch.bmm(a3, bp, out_dtype=torch.float32)
        y = torch.baddbmm(y, a2, bp[:, :2 * k], out_dtype=torch.float32)
        return torch.baddbmm(y, a1, bp[:, :k], out_dtype=torch.float32)
    try:
        ys = split_gemm()
        ts = bench(split_gemm)
        e32 = float((y32.double() - ref).abs().max() / ref.abs().max())
        es = float((ys.double() - ref).abs().max() / ref.abs().max())
        tsplit = bench(lambda: torch.cat(pieces(x), 1))
        print(f'b{b} {m}x{k}x{n}: f32 bmm {t32:7.1f} us (err {e32:.1e}) | 3 bf16 GEMMs {ts:7.1f} us (err {es:.1e}) | torch split of x {tsplit:7.1f} us')
    except Exception as e:
        print(f'b{b} {m}x{k}x{n}: f32 {t32:.1f} us; split failed: {type(e).__name__}: {str(e)[:200]}')
