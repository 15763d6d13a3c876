"""Per-block timing of the two routes through the first half of every MBConv block of HyperSeg-M's encoder at 1024x512:
(a) expand (MFMA kernel or library GEMM) + depthwise kernel, (b) the fused hs_mbconv_expand_dw_fwd launch.  Each route is
captured in a HIP graph of REP back-to-back calls and replayed; prints us per call and checks (b) against (a).
    python tools/bench_mbconv.py [H W]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hyperseg_amd import configs, functional as HF
from hyperseg_amd.utils.synthetic import fill_by_name
from hyperseg_amd.utils.inference import prepare_for_inference

REP = 20
torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 1024)
m = fill_by_name(configs.build('hyperseg-m').eval(), seed=0)
prepare_for_inference(m, fold_bn=False, fused_depthwise=True)
m = m.to(dev)
bb = m.backbone


def timed(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REP):
            out = fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * REP), out


x = torch.rand(1, 3, h, w, device=dev)
t = torch.nn.functional.silu(bb._bn0(bb._conv_stem(x)))
print(f'{"blk":>3} {"cin":>4} {"cmid":>5} {"k":>1} {"s":>1} {"in":>9} {"out":>9} {"unfused_us":>10} {"fused_us":>9} {"rel_err":>9}')
tot_a = tot_b = 0.0
for idx, blk in enumerate(bb._blocks):
    f = blk._fused_dw
    xin = t.contiguous()
    if f.expand is not None and xin.shape[1] <= 80:
        _, _, hh, ww = xin.shape
        ho = (hh + f.pad_h - f.k) // f.stride + 1
        wo = (ww + f.pad_w - f.k) // f.stride + 1

        def unfused():
            if not f.expand.uses_mfma(xin):
                r = f.expand.raw(xin)
                return HF.depthwise_conv_bn_act(r, blk._depthwise_conv.weight, f.stride, f.pad_t, f.pad_l, (ho, wo), f.scale,
                                                f.shift, act=3, pool=True, in_scale=f.expand.scale, in_shift=f.expand.shift)
            return HF.depthwise_conv_bn_act(f.expand(xin), blk._depthwise_conv.weight, f.stride, f.pad_t, f.pad_l, (ho, wo),
                                            f.scale, f.shift, act=3, pool=True)

        def fused():
            return HF.mbconv_expand_dw(xin, f.expand.conv.weight, f.expand.scale, f.expand.shift, blk._depthwise_conv.weight,
                                       f.stride, f.pad_t, f.pad_l, (ho, wo), f.scale, f.shift, pool=True)

        ta, (ya, pa) = timed(unfused)
        tb, (yb, pb) = timed(fused)
        err = float((ya - yb).abs().max() / ya.abs().max())
        perr = float((pa.sum(1) - pb.sum(1)).abs().max() / pa.sum(1).abs().max())
        tot_a += ta
        tot_b += tb
        print(f'{idx:3d} {xin.shape[1]:4d} {ya.shape[1]:5d} {f.k} {f.stride} {hh:4d}x{ww:<4d} {ho:4d}x{wo:<4d} {ta:10.2f} {tb:9.2f} '
              f'{err:9.1e} pool {perr:.1e}')
    t = blk(t)
print(f'sum unfused {tot_a:.1f} us   fused {tot_b:.1f} us')
if os.environ.get('HS_BENCH_MBCONV_FIRST_TABLE_ONLY'):
    sys.exit(0)

# ---- every 1x1 convolution of the encoder: library GEMM (torch.mm) vs hs_pointwise_conv_fwd (bare GEMM, no epilogue) ----
print(f'\n{"blk":>3} {"conv":>7} {"cin":>5} {"cout":>5} {"pixels":>7} {"mm_us":>8} {"mfma_us":>8}')
t = torch.nn.functional.silu(bb._bn0(bb._conv_stem(x)))
for idx, blk in enumerate(bb._blocks):
    f = blk._fused_dw
    xin = t.contiguous()
    convs = []
    if f.expand is not None:
        convs.append(('expand', blk._expand_conv.weight, xin))
    hh, ww = xin.shape[2:]
    ho = (hh + f.pad_h - f.k) // f.stride + 1
    wo = (ww + f.pad_w - f.k) // f.stride + 1
    convs.append(('project', blk._project_conv.weight, torch.randn(1, blk._project_conv.in_channels, ho, wo, device=dev)))
    for name, wt, inp in convs:
        cin, px = inp.shape[1], inp.shape[2] * inp.shape[3]
        if cin > 128:
            ta, _ = timed(lambda: torch.mm(wt.view(-1, cin), inp.view(cin, px)))
            print(f'{idx:3d} {name:>7} {cin:5d} {wt.shape[0]:5d} {px:7d} {ta:8.2f} {"-":>8}')
            continue
        ta, ya = timed(lambda: torch.mm(wt.view(-1, cin), inp.view(cin, px)))
        tb, yb = timed(lambda: HF.pointwise_conv(inp, wt, None, None, None, 0, None))
        err = float((ya.view(-1) - yb.view(-1)).abs().max() / ya.abs().max())
        print(f'{idx:3d} {name:>7} {cin:5d} {wt.shape[0]:5d} {px:7d} {ta:8.2f} {tb:8.2f}   err {err:.1e}')
    t = blk(t)
