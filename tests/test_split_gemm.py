"""hs_gemm_split_fwd (the encoder's 1x1 convolutions on the f16 matrix cores, include/hyperseg_hip.h).  The host side (weight
split, fragment order, plan) is checked on the CPU; the GPU tests run under plain ``-m gpu`` since round 3 (first GPU visit r5a:
16 passed; bench.py 1091 -> 1179 frames/s with the route on, profiles/round3_first_visit.txt)."""
import pytest
import torch

from conftest import G, rel_err

def test_split_weights_host_side():
    from hyperseg_amd import functional as HF
    from hyperseg_amd._hip import lib
    # plan: Cin rounded up to (2 | 4 | 8 waves) x (<= 5 k-steps) x 32; beyond 1280 the kernel declines
    assert [lib.hs_gemm_split_kp(k) for k in (1, 32, 80, 112, 192, 240, 480, 672, 1152, 1280)] == \
        [64, 64, 128, 128, 256, 256, 512, 768, 1280, 1280]
    # 1281 ... 2560: two chunks of <= 5 k-steps per wave -- the windowed 2x2 / stride-2 form only (hs_gemm_split_fwd declines)
    assert [lib.hs_gemm_split_kp(k) for k in (1281, 1600, 1920, 2560)] == [1536, 2048, 2048, 2560]
    assert lib.hs_gemm_split_kp(2561) < 0 and lib.hs_gemm_split_kp(0) < 0
    g = torch.Generator().manual_seed(0)
    for m, k in ((40, 240), (19, 80), (112, 672)):
        w = torch.randn(m, k, 1, 1, generator=g) * torch.logspace(-3, 3, m).view(m, 1, 1, 1)      # rows of very different size
        scale = torch.rand(m, generator=g) + 0.5
        sw = HF.gemm_split_weights(w, scale)
        assert (sw.c_out, sw.c_in, sw.kp) == (m, k, lib.hs_gemm_split_kp(k))
        rt, kst = -(-m // 16), sw.kp // 32
        assert tuple(sw.frag.shape) == (rt, kst, 2, 4, 16, 8) and sw.frag.dtype == torch.float16 and sw.inv.numel() == 16 * rt
        # back from fragment order: lane = row % 16 + 16 * kgroup holds w[16 R + row % 16][32 S + 8 kgroup + j]
        back = sw.frag.permute(2, 0, 4, 1, 3, 5).reshape(2, 16 * rt, sw.kp).float()
        ref = (w.flatten(1) * scale[:, None])
        rebuilt = (back[0].double() + back[1].double()) * sw.inv.double()[:, None]
        assert float(back[:, m:].abs().max() if 16 * rt > m else 0.0) == 0.0 and float(back[:, :, k:].abs().max()) == 0.0
        err = (rebuilt[:m, :k] - ref.double()).abs().amax(1) / ref.abs().amax(1).double()
        assert float(err.max()) < 2.0 ** -21                     # two f16 pieces of a row scaled to < 2^15
        assert float((back[0].abs().amax(1)[:m]).max()) < 2.0 ** 15
    assert HF.gemm_split_weights(torch.randn(8, 1920, generator=G(1034))) is None
    sw4 = HF.gemm_split_weights(torch.randn(8, 480, 2, 2, generator=G(1035)), max_k=2560)
    assert (sw4.c_in, sw4.kp) == (1920, 2048) and HF.gemm_split_weights(torch.randn(8, 641, 2, 2, generator=G(1036)), max_k=2560) is None
    with pytest.raises(Exception):
        HF.gemm_split(HF.gemm_split_weights(torch.randn(8, 64, generator=G(1037))), torch.rand(1, 64, 4, 4, generator=G(1038)))       # CPU tensors: no fallback


@pytest.mark.parametrize('m,k', [(40, 240), (112, 672), (192, 1152), (480, 80)])
def test_split_arithmetic_emulated(m, k):
    """The kernel's arithmetic restated with torch on the CPU from the REAL weight preparation: per wave K slice, the gated
    input column of every pixel is scaled by a power of two to < 2^15 and split into two f16 pieces; three f16 x f16 products
    (exact in f32) accumulate; column and row scales are undone at the end.  Against the float64 product the result must be at
    least as close as a plain f32 matmul (the claim DESIGN section 7.2 makes; the GPU probe measured 1.6-2.9e-7)."""
    from hyperseg_amd import functional as HF
    g = torch.Generator().manual_seed(m * k)
    n = 96
    w = torch.randn(m, k, generator=g) / k ** 0.5
    x = torch.randn(k, n, generator=g) * torch.rand(k, 1, generator=g) * 4
    gate = torch.rand(k, generator=g)
    sw = HF.gemm_split_weights(w)
    rt = sw.frag.shape[0]
    back = sw.frag.permute(2, 0, 4, 1, 3, 5).reshape(2, 16 * rt, sw.kp)                    # [piece][row][k], f16
    steps = sw.kp // 32
    nwv = 2
    while nwv < 8 and nwv < steps:
        nwv *= 2
    ks = steps // nwv
    assert nwv * ks * 32 == sw.kp
    xg = torch.nn.functional.pad(x * gate[:, None], (0, 0, 0, sw.kp - k))
    y = torch.zeros(16 * rt, n)
    for wave in range(nwv):
        sl = slice(wave * ks * 32, (wave + 1) * ks * 32)
        eb = (xg[sl].abs().amax(0).contiguous().view(torch.int32) >> 23).clamp(27, 254)
        sc, inv = torch.ldexp(torch.ones(n), 141 - eb), torch.ldexp(torch.ones(n), eb - 141)
        xs = xg[sl] * sc[None, :]
        bh = xs.half()
        bl = (xs - bh.float()).half()
        ah, al = back[0][:, sl].double(), back[1][:, sl].double()
        part = al @ bh.double() + ah @ bl.double() + ah @ bh.double()                     # products of f16 pairs: exact
        y += part.float() * inv[None, :]
    y = (y * sw.inv[:, None])[:m]
    ref = w.double() @ (x.double() * gate.double()[:, None])
    err_split = float((y.double() - ref).abs().max() / ref.abs().max())
    err_f32 = float(((w @ (x * gate[:, None])).double() - ref).abs().max() / ref.abs().max())
    assert err_split < 5e-7 and err_split < 2 * err_f32 + 1e-7


@pytest.mark.gpu
@pytest.mark.parametrize('m,k,hw,batch', [(40, 240, (64, 128), 1), (80, 480, (32, 64), 2), (112, 672, (32, 64), 1), (192, 1152, (16, 32), 1),
                                          (320, 1152, (16, 32), 2), (480, 80, (32, 64), 1), (19, 33, (5, 7), 1), (1280, 320, (16, 32), 1),
                                          (40, 36, (33, 129), 1),      # odd pixel count on a wide grid: the scalar-load form, 32-pixel blocks
                                          # round 6: 1280 < K <= 2560 on the two-chunk form (HyperSeg-M's last project conv: K = 1920)
                                          (320, 1920, (16, 32), 1), (320, 1920, (16, 32), 2), (100, 1400, (8, 16), 1), (64, 2560, (8, 8), 1)])
def test_gemm_split_vs_float64(m, k, hw, batch):
    from hyperseg_amd import functional as HF
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(m + k)
    w = (torch.randn(m, k, generator=g) / k ** 0.5).to(dev)
    x = (torch.randn(batch, k, *hw, generator=g) * torch.rand(1, k, 1, 1, generator=g) * 4).to(dev)
    gate = torch.rand(batch, k, generator=g).to(dev)
    scale = (torch.rand(m, generator=g) + 0.5).to(dev)
    sw = HF.gemm_split_weights(w, scale, max_k=2560)
    ref = torch.einsum('ok,bkn->bon', (w * scale[:, None]).double(), (x.flatten(2) * gate[:, :, None]).double()).view(batch, m, *hw)
    y = HF.gemm_split(sw, x, gate=gate)
    assert rel_err(y.double().cpu(), ref.cpu()) < 2e-6
    y0 = HF.gemm_split(HF.gemm_split_weights(w, max_k=2560), x)
    ref0 = torch.einsum('ok,bkn->bon', w.double(), x.flatten(2).double()).view(batch, m, *hw)
    assert rel_err(y0.double().cpu(), ref0.cpu()) < 2e-6
    acc = torch.ones_like(y)
    assert HF.gemm_split(sw, x, gate=gate, residual=acc, out=acc) is acc                   # in place onto a skip tensor
    assert rel_err(acc.double().cpu(), (ref + 1).cpu()) < 2e-6
    shift = torch.randn(m, generator=g).to(dev)
    res = torch.randn(batch, m, *hw, generator=g).to(dev)
    for act, fn in ((0, lambda t: t), (1, torch.relu), (2, lambda t: t.clamp(0, 6)), (3, lambda t: t * torch.sigmoid(t))):
        ya = HF.gemm_split(sw, x, gate=gate, shift=shift, act=act, residual=res)
        want = fn(ref + shift.double().view(1, -1, 1, 1)) + res.double()
        assert rel_err(ya.double().cpu(), want.cpu()) < (2e-5 if act == 3 else 2e-6)       # swish: the fast exp2 / rcp form


@pytest.mark.gpu
@pytest.mark.parametrize('m,c,hw,batch', [(640, 640, (16, 32), 1), (96, 64, (8, 8), 2), (130, 250, (6, 12), 1), (320, 320, (32, 64), 1),
                                          (48, 400, (4, 4), 1)])
def test_gemm_split_conv2x2_vs_float64(m, c, hw, batch):
    """Conv2d(c, m, kernel 2, stride 2) + shift + ReLU with the window read on load (the context head's down blocks,
    hyperseg_v1_0.py:396-401; K = 4 c up to 2560 in two chunks per wave) against the float64 convolution."""
    import torch.nn.functional as F
    from hyperseg_amd import functional as HF
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(m + c)
    w = (torch.randn(m, c, 2, 2, generator=g) / (4 * c) ** 0.5).to(dev)
    x = (torch.randn(batch, c, *hw, generator=g) * torch.rand(1, c, 1, 1, generator=g) * 4).to(dev)
    scale = (torch.rand(m, generator=g) + 0.5).to(dev)
    shift = torch.randn(m, generator=g).to(dev)
    sw = HF.gemm_split_weights(w, scale, max_k=2560)
    assert sw is not None and sw.c_in == 4 * c
    ref = F.conv2d(x.double(), (w * scale.view(-1, 1, 1, 1)).double(), stride=2)
    y = HF.gemm_split_conv2x2(sw, x)
    assert y.shape == ref.shape and rel_err(y.double().cpu(), ref.cpu()) < 2e-6
    ya = HF.gemm_split_conv2x2(sw, x, shift=shift, act=HF.ACT_RELU)
    assert rel_err(ya.double().cpu(), torch.relu(ref + shift.double().view(1, -1, 1, 1)).cpu()) < 2e-6
    assert HF.gemm_split_weights(w, scale) is None or 4 * c <= 1280             # the 1x1 form stops at K = 1280
    # the global average of the result rides along as sums over blocks of 16 pixels; folded into a shift by hs_pooled_shift_fwd
    if batch == 1:
        p = (hw[0] // 2) * (hw[1] // 2)
        part = torch.full((1, m, -(-p // 16)), float('nan'), device=dev)
        yp = HF.gemm_split_conv2x2(sw, x, shift=shift, act=HF.ACT_RELU, pool_partial=part)
        assert torch.equal(yp, ya)
        mean = yp.double().mean((2, 3)).view(m)
        assert rel_err(part.double().sum(2).view(m).cpu() / p, mean.cpu()) < 1e-6
        wb = torch.randn(24, m, generator=g).to(dev)
        base = torch.randn(24, generator=g).to(dev)
        got = HF.pooled_shift(part, p, wb, base)
        assert rel_err(got.double().cpu(), (base.double() + wb.double() @ mean).cpu()) < 1e-6
    with pytest.raises(ValueError):
        HF.gemm_split_conv2x2(sw, x[:, :, :, :hw[1] - 2].contiguous())          # width must be a multiple of 4


@pytest.mark.gpu
@pytest.mark.parametrize('m,k,hw,batch', [(640, 640, (8, 16), 1), (40, 96, (5, 6), 2), (320, 1152, (16, 32), 1)])
def test_gemm_split_up2_vs_float64(m, k, hw, batch):
    """1x1 conv + shift + ReLU stored nearest-2x upsampled (hs_gemm_split_up2_fwd), into a channel slice of a larger tensor."""
    import torch.nn.functional as F
    from hyperseg_amd import functional as HF
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(m * k)
    w = (torch.randn(m, k, generator=g) / k ** 0.5).to(dev)
    x = torch.randn(batch, k, *hw, generator=g).to(dev)
    shift = torch.randn(m, generator=g).to(dev)
    sw = HF.gemm_split_weights(w)
    ref = F.interpolate(torch.relu(torch.einsum('ok,bkn->bon', w.double(), x.flatten(2).double()).view(batch, m, *hw)
                                   + shift.double().view(1, -1, 1, 1)), scale_factor=2, mode='nearest')
    y = HF.gemm_split_up2(sw, x, shift=shift, act=HF.ACT_RELU)
    assert y.shape == ref.shape and rel_err(y.double().cpu(), ref.cpu()) < 2e-6
    if batch == 1:
        big = torch.zeros(1, m + 8, 2 * hw[0], 2 * hw[1], device=dev)
        HF.gemm_split_up2(sw, x, shift=shift, act=HF.ACT_RELU, out=big[:, 8:])
        assert torch.equal(big[:, 8:], y) and float(big[:, :8].abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize('batch,size', [(1, (256, 512)), (2, (256, 256)), (1, (512, 1024))])
def test_prepared_model_with_split_gemm(batch, size):
    import copy
    from hyperseg_amd import configs
    from hyperseg_amd.utils.inference import prepare_for_inference
    from hyperseg_amd.utils.synthetic import fill_by_name
    dev = torch.device('cuda:0')
    stock = fill_by_name(configs.build('hyperseg-m').eval(), seed=9)
    fused = copy.deepcopy(stock)
    prepare_for_inference(fused, fold_bn=False, fused_depthwise=True, split_gemm=True)
    stock, fused = stock.to(dev), fused.to(dev)
    x = torch.rand(batch, 3, *size, generator=G(1039)).to(dev)
    with torch.no_grad():
        for a, b in zip(stock.backbone(x), fused.backbone(x)):
            assert rel_err(b.cpu(), a.cpu()) < 5e-5
        ys = stock(x).cpu()
        assert rel_err(fused(x).cpu(), ys) < 1e-4
        assert rel_err(fused(x).cpu(), ys) < 1e-4          # twice: the in-place skip accumulation
