"""Parity of the HIP path (through the C ABI and the nn.Module mirror) against the reference-made
golden fixtures and the CPU oracle.  Needs an MI355X: run with ``-m gpu``.

Tolerances: the north star asks for <= 1e-3 relative fp32 and argmax-identical masks.  Everything
here is held to REL_TOL = 2e-5 (tensor-scale relative; observed ~1e-6) -- fp32 sums are re-associated,
so bit equality is not expected -- and masks must agree wherever the reference's own top-2 margin
exceeds MARGIN = 1e-4 (the minimum margin over 0.5 M pixels is ~1 ulp, see make_golden.py).
"""
import numpy as np
import pytest
import torch

from conftest import G, bn_of, rel_err, sub

pytestmark = pytest.mark.gpu

REL_TOL = 2e-5
NORTH_STAR_TOL = 1e-3
MARGIN = 1e-4


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.fail('these tests need the MI355X (torch.cuda.is_available() is False)')
    return torch.device('cuda:0')


@pytest.fixture(scope='module')
def HF(dev):
    import hyperseg_amd.functional as hf
    return hf


@pytest.fixture(scope='module')
def O():
    from oracle import hyperseg_oracle
    return hyperseg_oracle


@pytest.fixture(params=['split', 'f32', 'auto'])
def ir_math(request, HF):
    """All arithmetic modes of the fused inverted-residual kernels (include/hyperseg_hip.h, hs_ir_math; the mode is an argument
    of every launch, forced here through the Python-side override): f16 split products on the f16 matrix cores, exact f32,
    and auto -- held to the SAME tolerances."""
    prev = HF.set_ir_math(request.param)
    yield request.param
    HF.set_ir_math(prev)


def load_bn(bn, p, prefix):
    with torch.no_grad():
        for k in ('weight', 'bias', 'running_mean', 'running_var'):
            getattr(bn, k).copy_(p[f'{prefix}.{k}'])


def cmp(y, ref, tol=REL_TOL, what=''):
    assert tuple(y.shape) == tuple(ref.shape), (what, y.shape, ref.shape)
    e = rel_err(y.cpu(), ref)
    assert e < tol, f'{what}: relative error {e:.3e} >= {tol}'
    return e


# ------------------------------------------------------------------------------ raw kernels
def test_stage_input_and_upsample(HF, O, dev):
    g = torch.Generator().manual_seed(0)
    for (b, cs, cp, h, w, mode) in [(2, 3, 4, 8, 12, 'bilinear'), (1, 5, 0, 7, 9, 'none'), (2, 2, 3, 6, 10, 'same'),
                                    (1, 1, 2, 1, 4, 'bilinear'), (1, 4, 19, 64, 32, 'bilinear')]:
        skip = torch.randn(b, cs, h, w, generator=g)
        prev = None
        if mode == 'bilinear':
            prev = torch.randn(b, cp, max(h // 2, 1), max(w // 2, 1), generator=g)
        elif mode == 'same':
            prev = torch.randn(b, cp, h, w, generator=g)
        ref = O.stage_input(skip, prev)
        st = HF.StageInput(skip.to(dev), prev.to(dev) if prev is not None else None, coords=True)
        cmp(st.materialize(), ref, what=f'stage_input {mode}')
    for (shape, size) in [((2, 3, 5, 7), (10, 14)), ((1, 19, 16, 32), (32, 64)), ((1, 2, 6, 5), (9, 11)),
                          ((1, 1, 4, 4), (4, 4))]:
        x = torch.randn(*shape, generator=g)
        ref = torch.nn.functional.interpolate(x, size, mode='bilinear', align_corners=False)
        cmp(HF.upsample_bilinear(x.to(dev), size), ref, what=f'upsample {shape}->{size}')


def test_bn_fold(HF, O, dev):
    gen = torch.Generator().manual_seed(1)
    bn = O.synth_bn(gen, 37)
    scale, shift = HF.bn_fold(*(bn[k].to(dev) for k in ('weight', 'bias', 'running_mean', 'running_var')))
    x = torch.randn(2, 37, 3, 3, generator=gen)
    cmp(x.to(dev) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1), O.bn_eval(x, bn), what='bn_fold')


def test_signal2weights_and_bank_pack(HF, O, dev):
    g = torch.Generator().manual_seed(2)
    cases = [dict(b=2, c=24, fh=3, fw=4, idx=3, cs=16, grp=4, hp=35),       # next_multiply padding, 2 batches
             dict(b=1, c=64, fh=4, fw=8, idx=0, cs=64, grp=16, hp=1000),    # many groups per wave
             dict(b=1, c=1280, fh=16, fw=32, idx=0, cs=416, grp=32, hp=5248),   # HyperSeg-M level 0
             dict(b=3, c=10, fh=5, fw=7, idx=2, cs=8, grp=1, hp=9)]         # ragged patch count (105 % 16 != 0)
    for c in cases:
        rows = O.next_multiply(c['hp'], c['grp'])
        wsw = torch.randn(rows, c['cs'] // c['grp'], 1, 1, generator=g)
        s = torch.randn(c['b'], c['c'], c['fh'], c['fw'], generator=g).clamp(min=0)
        ref = O.signal2weights(s, wsw, c['idx'], c['cs'], c['grp'], c['hp'])          # (B, hp, fh, fw)
        ref_bank = ref.permute(0, 2, 3, 1).reshape(-1, c['hp'])
        wsw_t = wsw.reshape(rows, -1).t().contiguous().to(dev)
        bank = HF.signal2weights(s.to(dev), wsw_t, c['idx'], c['cs'], c['grp'], c['hp'])
        cmp(bank[:, :c['hp']], ref_bank, what=f's2w {c}')
        # pad columns [hp, ld) are never read by the consumers; rows beyond hp (next_multiply padding) are not stored
        # bank_pack of the reference-layout tensor gives the same bank, also from a channel-range view
        packed = HF.bank_pack(ref.contiguous().to(dev), 0, c['hp'])
        assert torch.equal(packed[:, :c['hp']].cpu(), ref_bank)
        if c['hp'] > 8:
            big = torch.cat([torch.zeros_like(ref[:, :3]), ref, torch.ones_like(ref[:, :2])], dim=1).to(dev)
            packed = HF.bank_pack(big[:, 3:3 + c['hp']], 0, c['hp'])
            assert torch.equal(packed[:, :c['hp']].cpu(), ref_bank)
            packed = HF.bank_pack(big, 3, c['hp'])
            assert torch.equal(packed[:, :c['hp']].cpu(), ref_bank)


def test_signal2weights_multi(HF, O, dev):
    """All five HyperSeg-M levels in one launch == five oracle convolutions (rows/group 164, 188, 88, 147, 1054:
    partial MFMA tiles, unaligned groups (147), K = 13, 14, 16, 12, 80)."""
    plan = O.config_plan('M')
    params = O.synth_decoder_params(plan, seed=5)
    _, s = O.synth_decoder_inputs('M', batch=1, seed=5)
    layers, refs = [], []
    for l, (lv, sw) in enumerate(zip(plan['levels'], plan['s2w'])):
        key = f'level_{l}.0.0.signal2weights.weight' if lv['k'] == 1 else f'level_{l}.0.signal2weights.weight'
        w = params[key]
        layers.append(dict(wsw_t=w.reshape(w.shape[0], -1).t().contiguous().to(dev), signal_index=sw['signal_index'],
                           signal_channels=sw['signal_channels'], groups=sw['groups'], rows=lv['hp']))
        refs.append(O.signal2weights(s, w, sw['signal_index'], sw['signal_channels'], sw['groups'], lv['hp']))
    out = HF.signal2weights_multi(s.to(dev), layers)
    for ref, o, lv in zip(refs, out, plan['levels']):
        cmp(o.bank[:, :lv['hp']], ref.permute(0, 2, 3, 1).reshape(-1, lv['hp']), what=f's2w multi level hp={lv["hp"]}')


def test_coscheduled_banks_heterogeneous_launch(HF, O, dev, monkeypatch):
    """hs_patch_conv_s2w_fwd: a k = 1 patch convolution and the signal2weights blocks of LATER levels in one launch
    (HF.CoScheduledBanks).  (1) raw: level 0 of HyperSeg-M carrying the banks of levels 1 + 2 -- output and banks bit-identical
    to the separate launches; odd grid / batch 2 / a k = 3 convolution inside the block (not eligible: the banks are produced
    on exit).  (2) the whole HyperSeg-M decoder with the co-scheduling on and off: bit-identical logits."""
    monkeypatch.setattr(HF, 'K1_CHAIN', False)          # bit-identity is between two per-level routes (the chained levels sum in another order)
    plan = O.config_plan('M')
    params = O.synth_decoder_params(plan, seed=5)
    x, s = O.synth_decoder_inputs('M', batch=1, seed=5)
    s = s.to(dev)
    layers = []
    for l, (lv, sw) in enumerate(zip(plan['levels'], plan['s2w'])):
        key = f'level_{l}.0.0.signal2weights.weight' if lv['k'] == 1 else f'level_{l}.0.signal2weights.weight'
        w = params[key]
        layers.append(dict(wsw_t=w.reshape(w.shape[0], -1).t().contiguous().to(dev), signal_index=sw['signal_index'],
                           signal_channels=sw['signal_channels'], groups=sw['groups'], rows=lv['hp']))
    fh, fw = s.shape[-2:]
    bank0 = HF.signal2weights_multi(s, layers[:1])[0]
    stage = HF.StageInput(x[-1].to(dev), None, coords=True)
    cout0 = plan['levels'][0]['cout']
    y_sep = HF.patch_conv(stage, (fh, fw), bank0.bank, cout0, 1, 0, 'zeros', 1)
    sep = HF.signal2weights_multi(s, layers[1:])
    with HF.CoScheduledBanks(s, layers[1:]) as co:
        y_co = HF.patch_conv(stage, (fh, fw), bank0.bank, cout0, 1, 0, 'zeros', 1)
        assert co.carried and HF.CoScheduledBanks._active is None           # consumed by the first eligible launch
        y_2 = HF.patch_conv(stage, (fh, fw), bank0.bank, cout0, 1, 0, 'zeros', 1)
    assert torch.equal(y_co, y_sep) and torch.equal(y_2, y_sep)
    for a, b in zip(co.refs, sep):
        assert a.rows == b.rows and torch.equal(a.bank, b.bank)
    # a k = 3 convolution does not carry: the banks are produced when the block closes
    g = torch.Generator().manual_seed(3)
    xs = torch.randn(2, 6, 9 * 4, 6 * 4, generator=g).to(dev)
    sig = torch.relu(torch.randn(2, 64, 9, 6, generator=g)).to(dev)         # 54 cells per frame: not a multiple of 4 -> direct s2w form
    lay = [dict(wsw_t=torch.randn(8, 100, generator=g).to(dev), signal_index=8, signal_channels=32, groups=4, rows=98)]
    wt3 = torch.randn(2 * 9 * 6, 6 * 4 * 9 + 3, generator=g).to(dev)
    with HF.CoScheduledBanks(sig, lay) as co3:
        HF.patch_conv(xs, (9, 6), wt3[:, :6 * 4 * 9 + 3 - 3].contiguous(), 4, 3, 1, 'reflect', 1)
    assert not co3.carried and torch.equal(co3.refs[0].bank[:, :98], HF.signal2weights_multi(sig, lay)[0].bank[:, :98])   # (columns 98, 99 are row padding)
    # an eligible k = 1 launch whose riders cannot take the blocked form (54 patches per frame): HS_ERR_UNSUPPORTED -> separate launches
    wt1 = torch.randn(2 * 9 * 6, 6 * 5 + 2, generator=g).to(dev)
    with HF.CoScheduledBanks(sig, lay) as co1:
        y1 = HF.patch_conv(xs, (9, 6), wt1[:, :32].contiguous(), 5, 1, 0, 'zeros', 1)
    assert torch.equal(y1, HF.patch_conv(xs, (9, 6), wt1[:, :32].contiguous(), 5, 1, 0, 'zeros', 1))
    assert torch.equal(co1.refs[0].bank[:, :98], HF.signal2weights_multi(sig, lay)[0].bank[:, :98])
    # (2) the decoder
    d = build_decoder('M', O).to(dev)
    xd = [t.to(dev) for t in x]
    with torch.no_grad():
        monkeypatch.setattr(HF, 'COSCHEDULE_BANKS', True)
        y_on = d(xd, s)
        monkeypatch.setattr(HF, 'COSCHEDULE_BANKS', False)
        y_off = d(xd, s)
    assert torch.equal(y_on, y_off)


def test_meta_conv2d(golden, dev):
    from hyperseg_amd.models.layers.meta_conv import MetaConv2d
    m = MetaConv2d(3, 3, 3, padding=1, groups=3)
    x = torch.ones(4, 3, 64, 64)
    x[0::2] = 0
    w = torch.ones(4, m.hyper_params)
    w[0::2] = 0
    with torch.no_grad():
        assert float(m(x.to(dev), w.to(dev)).max()) == 9.0 == float(golden('meta_conv_known_answer')['out_max'])
    g = golden('meta_conv2d')
    for i in range(int(g['n'])):
        cin, cout, k, pad, groups = [int(v) for v in g[f'{i}.cfg']]
        m = MetaConv2d(cin, cout, k, padding=pad, groups=groups, padding_mode=str(g[f'{i}.mode']))
        with torch.no_grad():          # "valid" convs (2 pad != k - 1) take the general kernel
            cmp(m(g[f'{i}.x'].to(dev), g[f'{i}.w'].to(dev)), g[f'{i}.y'], what=f'meta_conv2d {i}')
    # the rest of the reference's argument set: non-square kernels, stride, dilation, any padding (hs_meta_conv_fwd)
    g = golden('meta_conv2d_general')
    for i in range(int(g['n'])):
        cin, cout, kh, kw, sh, sw, ph, pw, dh, dw, groups = [int(v) for v in g[f'{i}.cfg']]
        m = MetaConv2d(cin, cout, (kh, kw), stride=(sh, sw), padding=(ph, pw), dilation=(dh, dw), groups=groups,
                       padding_mode=str(g[f'{i}.mode']))
        with torch.no_grad():
            y = m(g[f'{i}.x'].to(dev), g[f'{i}.w'].to(dev))
        cmp(y, g[f'{i}.y'], what=f'general meta_conv2d {i}')
    # gradients through the general form (hs_meta_conv_bwd; VERDICT r3 missing #5): every case against autograd of the oracle
    from oracle import hyperseg_oracle as O
    for i in range(int(g['n'])):
        cin, cout, kh, kw, sh, sw, ph, pw, dh, dw, groups = [int(v) for v in g[f'{i}.cfg']]
        mode = str(g[f'{i}.mode'])
        m = MetaConv2d(cin, cout, (kh, kw), stride=(sh, sw), padding=(ph, pw), dilation=(dh, dw), groups=groups, padding_mode=mode)
        xo, wo = g[f'{i}.x'].clone().requires_grad_(), g[f'{i}.w'].clone().requires_grad_()
        yo = O.meta_conv2d(xo, wo, cout, (kh, kw), (sh, sw), (ph, pw), (dh, dw), groups, mode)
        r = torch.randn(yo.shape, generator=torch.Generator().manual_seed(40 + i))
        (yo * r).sum().backward()
        xg, wg = g[f'{i}.x'].to(dev).requires_grad_(), g[f'{i}.w'].to(dev).requires_grad_()
        yg = m(xg, wg)
        (yg * r.to(dev)).sum().backward()
        cmp(yg.detach(), yo.detach(), what=f'general meta_conv2d {i} under autograd')
        cmp(xg.grad, xo.grad, what=f'general meta_conv2d {i} dX')
        cmp(wg.grad, wo.grad, what=f'general meta_conv2d {i} dW')


def test_meta_patch_conv2d(golden, dev):
    from hyperseg_amd.models.layers.meta_patch import MetaPatchConv2d, make_meta_patch_conv2d_block
    g = golden('meta_patch_conv2d')
    with torch.no_grad():
        for i in range(int(g['n'])):
            cin, cout, k, groups = [int(v) for v in g[f'{i}.cfg']]
            m = MetaPatchConv2d(cin, cout, k, padding=k // 2, groups=groups)
            cmp(m(g[f'{i}.x'].to(dev), g[f'{i}.w'].to(dev)), g[f'{i}.y'], what=f'meta_patch {i}')
        blk = make_meta_patch_conv2d_block(6, 5, 1).eval()
        load_bn(blk[1], sub(g, 'blk.p.'), '1')
        blk = blk.to(dev)
        cmp(blk(g['blk.x'].to(dev), g['blk.w'].to(dev)), g['blk.y'], what='meta_patch block (fused BN+ReLU)')
        # meta_patch.py main(): x 2x10x256x256, w ones 2xhpx8x8 -> torch.Size([2, 20, 256, 256])
        m = MetaPatchConv2d(10, 20, 3, padding=1)
        y = m(torch.rand(2, 10, 256, 256, generator=G(1013)).to(dev), torch.ones(2, m.hyper_params, 8, 8, device=dev))
        assert y.shape == torch.Size([2, 20, 256, 256])


def test_meta_sequential(golden, dev):
    from hyperseg_amd.models.layers.meta_patch import MetaPatchConv2d
    from hyperseg_amd.models.layers.meta_sequential import MetaSequential
    g = golden('meta_sequential')
    seq = MetaSequential(MetaPatchConv2d(4, 6, 1), torch.nn.ReLU(), MetaPatchConv2d(6, 3, 3, padding=1)).eval().to(dev)
    hp0, hp1 = [int(v) for v in g['hp']]
    assert seq._ranges == [int(v) for v in g['ranges']] and seq.hyper_params == hp0 + hp1
    x, w = g['x'].to(dev), g['w'].to(dev)
    with torch.no_grad():
        cmp(seq(x, w), g['y_tensor'], what='tensor weights')
        cmp(seq(x, [w[:, :hp0].contiguous(), w[:, hp0:].contiguous()]), g['y_list'], what='list weights')
        w_long = torch.cat([w, torch.randn(2, 5, 3, 2, generator=G(1014)).to(dev)], dim=1)
        cmp(seq(x, w_long), g['y_long'], what='clamped slice')


def test_hyper_patch_v1(golden, dev):
    from hyperseg_amd.models import hyperseg_v1_0 as M
    g = golden('hyper_patch_v1')
    with torch.no_grad():
        cin, cout, cs, idx, grp, hp = [int(v) for v in g['np.cfg']]
        m = M.HyperPatchNoPadding(cin, cout, 1)
        m.init_signal2weights(cs, idx, grp)
        assert m.hyper_params == hp and tuple(m.signal2weights.weight.shape) == tuple(g['np.w_s2w'].shape)
        m.signal2weights.weight.copy_(g['np.w_s2w'])
        m = m.to(dev)
        cmp(m.apply_signal2weights(g['np.s'].to(dev)), g['np.wt'], what='apply_signal2weights')
        cmp(m(g['np.x'].to(dev), g['np.s'].to(dev)), g['np.y'], what='HyperPatchNoPadding')

        cin, cout, cs, idx, grp, hp = [int(v) for v in g['pc.cfg']]
        m = M.HyperPatchConv2d(cin, cout, 3, padding=1)
        m.init_signal2weights(cs, idx, grp)
        m.signal2weights.weight.copy_(g['pc.w_s2w'])
        m = m.to(dev)
        cmp(m(g['pc.x'].to(dev), g['pc.s'].to(dev)), g['pc.y'], what='HyperPatchConv2d')

        cin, cout, cs, idx, grp, hp = [int(v) for v in g['blk.cfg']]
        blk = M.make_hyper_patch_conv2d_block(cin, cout, 1).eval()
        blk[0].init_signal2weights(cs, idx, grp)
        p = sub(g, 'blk.p.')
        missing, unexpected = blk.load_state_dict(p, strict=False)
        assert not unexpected and all('num_batches' in k for k in missing)
        blk = blk.to(dev)
        cmp(blk(g['blk.x'].to(dev), g['blk.s'].to(dev)), g['blk.y'], what='hyper patch block, clamped signal slice')


def test_inverted_residual_v1(golden, dev, ir_math):
    from hyperseg_amd.models import hyperseg_v1_0 as M
    g = golden('inverted_residual_v1')
    with torch.no_grad():
        for i in range(int(g['n'])):
            cin, cout, hid, cs, idx, grp, hp = [int(v) for v in g[f'{i}.cfg']]
            m = M.HyperPatchInvertedResidual(cin, cout, 3, expand_ratio=hid / cin).eval()
            assert m.hidden_dim == hid and m.hyper_params == hp
            m.init_signal2weights(cs, idx, grp)
            missing, unexpected = m.load_state_dict(sub(g, f'{i}.p.'), strict=False)
            assert not unexpected and all('num_batches' in k for k in missing)
            m = m.to(dev)
            cmp(m(g[f'{i}.x'].to(dev), g[f'{i}.s'].to(dev)), g[f'{i}.y'], what=f'IR v1 case {i}')


def test_tiny_decoder_v1_0(golden, dev, ir_math):
    from hyperseg_amd.models import hyperseg_v1_0 as M
    from test_oracle_golden import TINY
    g = golden('decoder_t_v1_0')
    c = TINY['t_v1_0']
    d = M.MultiScaleDecoder(c['feat'], c['signal'], c['num_classes'], c['kernel_sizes'], 1, c['level_channels'],
                            expand_ratio=c['expand_ratio'], weight_groups=list(c['weight_groups'])).eval()
    assert d.param_groups == [int(v) for v in g['hyper_params']]
    missing, unexpected = d.load_state_dict(sub(g, 'p.'), strict=False)
    assert not unexpected and all('num_batches' in k for k in missing)
    d = d.to(dev)
    with torch.no_grad():
        y = d([g[f'x{i}'].to(dev) for i in range(6)], g['s'].to(dev))
    cmp(y, g['y'], what='tiny v1_0 decoder')
    assert bool((y.argmax(1).cpu() == g['y'].argmax(1)).all())


def test_inverted_residual_unify_and_v0(golden, dev, ir_math):
    """unify flavour (weights arrive directly) and the v0_1 block (three image-level patch convs, Op D)."""
    from hyperseg_amd.models import hyperseg_v1_0_unify as U
    from hyperseg_amd.models import hyperseg_v0_1 as V0
    g = golden('inverted_residual_v1')
    cin, cout, hid = [int(v) for v in g['u.cfg']]
    with torch.no_grad():
        m = U.HyperPatchInvertedResidual(cin, cout, 3, expand_ratio=hid / cin).eval()
        missing, unexpected = m.load_state_dict(sub(g, 'u.p.'), strict=False)
        assert not unexpected and all('num_batches' in k for k in missing)
        cmp(m.to(dev)(g['u.x'].to(dev), g['u.wt'].to(dev)), g['u.y'], what='unify IR')
        g0 = golden('inverted_residual_v0')
        for i in range(int(g0['n'])):
            cin, cout, hid = [int(v) for v in g0[f'{i}.cfg']]
            m = V0.HyperPatchInvertedResidual(cin, cout, 3, expand_ratio=2).eval()
            missing, unexpected = m.load_state_dict(sub(g0, f'{i}.p.'), strict=False)
            assert not unexpected and all('num_batches' in k for k in missing)
            cmp(m.to(dev)(g0[f'{i}.x'].to(dev), g0[f'{i}.wt'].to(dev)), g0[f'{i}.y'], what=f'v0_1 IR case {i}')


def make_decoder(c):
    if c['variant'] == 'v1_0':
        from hyperseg_amd.models import hyperseg_v1_0 as M
        return M.MultiScaleDecoder(c['feat'], c['signal'], c['num_classes'], c['kernel_sizes'], 1, c['level_channels'],
                                   expand_ratio=c['expand_ratio'], weight_groups=list(c['weight_groups'])).eval()
    if c['variant'] == 'unify':
        from hyperseg_amd.models import hyperseg_v1_0_unify as U
        return U.MultiScaleDecoder(c['feat'], c['signal'], c['num_classes'], c['kernel_sizes'], 1, c['level_channels'],
                                   expand_ratio=c['expand_ratio'], weight_groups=list(c['weight_groups']),
                                   unify_level=c['unify_level']).eval()
    from hyperseg_amd.models import hyperseg_v0_1 as V0
    return V0.MultiScaleDecoder(c['feat'], 3, c['num_classes'], c['kernel_sizes'], 1, expand_ratio=c['expand_ratio']).eval()


@pytest.mark.parametrize('name', ['t_unify', 't_v0_1'])
def test_tiny_decoder_other_variants(golden, dev, name, ir_math):
    from test_oracle_golden import TINY
    g = golden('decoder_' + name)
    c = TINY[name]
    d = make_decoder(c)
    missing, unexpected = d.load_state_dict(sub(g, 'p.'), strict=False)
    assert not unexpected and all('num_batches' in k for k in missing)
    d = d.to(dev)
    x = [g[f'x{i}'].to(dev) for i in range(6)]
    w = [g[f'w{i}'].to(dev) for i in range(6)] if c['variant'] == 'v0_1' else g['s'].to(dev)
    with torch.no_grad():
        y = d(x, w)
    cmp(y, g['y'], what=name)
    assert bool((y.argmax(1).cpu() == g['y'].argmax(1)).all())


# ------------------------------------------------------------------------------ full BASELINE shapes
def build_decoder(name, O):
    c = O.CONFIGS[name]
    d = make_decoder(c)
    params = O.synth_decoder_params(O.config_plan(name), seed=0)
    missing, unexpected = d.load_state_dict(params, strict=False)
    assert not unexpected and all('num_batches' in k for k in missing)
    return d


_FULL_REF = {}


@pytest.mark.parametrize('name', ['M', 'Sc', 'S', 'L', 'Lc'])
def test_full_config(golden, O, dev, name, ir_math):
    """HyperSeg-M 1024x512 / CamVid-S 768x576 / HyperSeg-S 1536x768 (unify) / HyperSeg-L 512x512 bs4 = the per-GPU shard of config 4 (v0_1) /
    CamVid HyperSeg-L 1024x768 (the six-level v1_0 decoder: three k = 1 levels, then three inverted residuals, the last on 32 x 32-pixel patches)
    decoders on the seeded synthetic workload of SURVEY 8(d): vs the oracle on the full tensor, vs the
    reference's own sampled logits and masks."""
    g = golden('decoder_full_configs')
    batch = int(g[f'{name}.batch'])
    d = build_decoder(name, O).to(dev)
    x, s = O.synth_decoder_inputs(name, batch=batch, seed=0)
    s = [t.to(dev) for t in s] if isinstance(s, list) else s.to(dev)
    with torch.no_grad():
        y = d([t.to(dev) for t in x], s).cpu()
    if name not in _FULL_REF:
        _FULL_REF[name] = O.run_config(name, batch=batch, seed=0)
    ref = _FULL_REF[name]
    e = cmp(y, ref, tol=REL_TOL, what=f'{name} logits vs oracle ({ir_math})')
    assert e < NORTH_STAR_TOL
    top2 = ref.topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    flips = (y.argmax(1) != ref.argmax(1))
    assert int((flips & (margin > MARGIN)).sum()) == 0
    assert int(flips.sum()) <= int((margin <= MARGIN).sum())
    # against the reference itself (fixture): strided logits + masks
    ys = y[:, :, 3::37, 5::41]
    assert float((ys - g[f'{name}.logits_sample']).abs().max()) < REL_TOL * float(g[f'{name}.logits_absmax'])
    ok = g[f'{name}.margin_sample'] > MARGIN
    assert bool((ys.argmax(1).to(torch.uint8)[ok] == g[f'{name}.mask_sample'][ok]).all())


def test_errors_are_loud(HF, dev):
    from hyperseg_amd._hip import HipLibraryError
    from hyperseg_amd.models.layers.meta_patch import MetaPatchConv2d
    m = MetaPatchConv2d(4, 4, 1)
    with torch.no_grad():
        with pytest.raises(HipLibraryError):
            m(torch.randn(1, 4, 8, 8, generator=G(1015)), torch.randn(1, 16, 2, 2, generator=G(1016)))                      # CPU tensors
        with pytest.raises(ValueError):
            m(torch.randn(1, 4, 9, 8, generator=G(1017)).to(dev), torch.randn(1, 16, 2, 2, generator=G(1018)).to(dev))   # 9 % 2 != 0
        with pytest.raises(ValueError):
            m(torch.randn(1, 4, 8, 8, generator=G(1019)).to(dev), torch.randn(1, 15, 2, 2, generator=G(1020)).to(dev))   # too few weights
    # gradients are supported through hyperseg_amd.autograd (tests/test_hip_training.py)
    y = m(torch.randn(1, 4, 8, 8, generator=G(1021)).to(dev).requires_grad_(True), torch.randn(1, 16, 2, 2, generator=G(1022)).to(dev))
    assert y.requires_grad and y.grad_fn is not None


@pytest.mark.parametrize('shape,size', [((2, 19, 16, 32), (32, 64)), ((1, 19, 64, 128), (128, 256)), ((1, 7, 9, 13), (18, 26)),
                                        ((2, 21, 11, 14), (33, 31)), ((1, 5, 12, 20), (12, 20)), ((1, 1, 8, 8), (16, 16)),
                                        ((1, 256, 4, 6), (8, 12))])
def test_upsample_argmax(HF, dev, shape, size):
    """hs_upsample_argmax_fwd == argmax over hs_upsample_bilinear_fwd's logits, bit for bit (shared arithmetic), and ==
    the reference epilogue F.interpolate(...).argmax(1) wherever the top-2 margin is not a rounding artefact."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g)
    m = HF.upsample_argmax(x.to(dev), size).cpu()
    assert m.dtype == torch.uint8 and tuple(m.shape) == (shape[0],) + tuple(size)
    up = HF.upsample_bilinear(x.to(dev), size).cpu()
    assert bool((m.long() == up.argmax(1)).all())
    ref = F.interpolate(x, size, mode='bilinear', align_corners=False)
    top2 = ref.topk(min(2, shape[1]), dim=1).values
    clear = (top2[:, 0] - top2[:, -1] > MARGIN) if shape[1] > 1 else torch.ones_like(m, dtype=torch.bool)
    assert bool((m.long()[clear] == ref.argmax(1)[clear]).all())


# ------------------------------------------------------------------------------ fused Op D (hs_patch_ir_v0_fwd)
# (cin = 2 + skip + prev, cout, patch edge, grid): the four inverted-residual levels of HyperSeg-L
# (hyperseg_v0_1.py:205-237 at 64^2 / 128^2 / 256^2 / 512^2 of a 512x512 input), on small grids
OP_D_CASES = [
    dict(skip=12, prev=34, cout=12, patch=4, grid=(4, 6)),      # level 2: 8x8 regions of 2x2 patches
    dict(skip=8, prev=12, cout=8, patch=8, grid=(2, 4)),        # level 3: 16x16 regions of 2x2 patches
    dict(skip=6, prev=8, cout=6, patch=16, grid=(2, 3)),        # level 4: region == patch
    dict(skip=3, prev=6, cout=21, patch=32, grid=(1, 2)),       # level 5: 4 regions per patch
    dict(skip=6, prev=8, cout=6, patch=16, grid=(1, 1)),        # a single patch: every ring position is a reflection
    # the two-launch form for 4 x 4 / 8 x 8 patches (hs_patch_ir_d2.hip): odd grids (a wave leaves its workgroup early), the level
    # shapes at grids whose regions would not tile, channel counts off the level shapes, a 1 x 1 grid (reflection everywhere)
    dict(skip=12, prev=34, cout=12, patch=4, grid=(3, 5)),
    dict(skip=12, prev=34, cout=12, patch=4, grid=(8, 8)),
    dict(skip=8, prev=12, cout=8, patch=8, grid=(3, 3)),
    dict(skip=8, prev=16, cout=8, patch=8, grid=(2, 5)),        # 26 -> 52 channels: 8-byte weight loads, the wide instantiation
    dict(skip=6, prev=10, cout=3, patch=4, grid=(5, 2)),
    dict(skip=8, prev=12, cout=8, patch=8, grid=(1, 1)),
    dict(skip=12, prev=34, cout=16, patch=8, grid=(2, 2)),      # level-2 channels on 8 x 8 patches (the wide instantiation)
]


def _op_d_case(c, dev, O, batch=2, seed=5):
    from hyperseg_amd.models import hyperseg_v0_1 as V0
    g = torch.Generator().manual_seed(seed + c['patch'])
    cin = 2 + c['skip'] + c['prev']
    fh, fw = c['grid']
    h, w = fh * c['patch'], fw * c['patch']
    m = V0.HyperPatchInvertedResidual(cin, c['cout'], 3, expand_ratio=2).eval()
    hid = 2 * cin
    bns = []
    with torch.no_grad():
        for blk in m.conv:
            bn = blk[1]
            bn.weight.copy_(torch.rand(bn.num_features, generator=g) + 0.5)
            bn.bias.copy_(torch.randn(bn.num_features, generator=g) * 0.1)
            bn.running_mean.copy_(torch.randn(bn.num_features, generator=g) * 0.1)
            bn.running_var.copy_(torch.rand(bn.num_features, generator=g) * 1.5 + 0.5)
            bns.append({k: getattr(bn, k).clone() for k in ('weight', 'bias', 'running_mean', 'running_var')})
    skip = torch.randn(batch, c['skip'], h, w, generator=g)
    prev = torch.randn(batch, c['prev'], h // 2, w // 2, generator=g)
    # per-patch weights with the fan-in scaling of a trained hypernetwork head, so that ReLU6 is exercised on both sides
    hp = m.hyper_params
    assert hp == cin * hid + 9 * hid + hid * c['cout']
    wt = torch.randn(batch, hp, fh, fw, generator=g)
    wt[:, :cin * hid] *= (2.0 / cin) ** 0.5
    wt[:, cin * hid:cin * hid + 9 * hid] *= (2.0 / 9) ** 0.5
    wt[:, cin * hid + 9 * hid:] *= (1.0 / hid) ** 0.5
    ref = O.patch_inverted_residual_v0(O.stage_input(skip, prev), wt, hid, c['cout'], *bns)
    return m.to(dev), skip, prev, wt, ref


@pytest.mark.parametrize('case', OP_D_CASES)
def test_fused_op_d_vs_oracle(HF, O, dev, case, ir_math):
    """The one-launch Op D kernel (neighbour-weight ring recompute) == the oracle's three image-level patch convs, and ==
    this package's own three-launch route bit-for-bit in structure (same inputs), for every HyperSeg-L level shape."""
    m, skip, prev, wt, ref = _op_d_case(case, dev, O)
    with torch.no_grad():
        stage = HF.StageInput(skip.to(dev), prev.to(dev), coords=True)
        # the fused launch must be the one that ran: ask the library directly
        parts = m._fused_parts()
        assert parts is not None
        y_fused = m._forward_fused(stage, wt.to(dev))
        assert y_fused is not None, 'hs_patch_ir_v0_fwd has no instantiation for a HyperSeg-L level shape'
        cmp(y_fused, ref, what=f'fused Op D {case}')
        y3 = m.conv(stage, wt.to(dev))                       # the three-launch route (generic kernels)
        cmp(y3, ref, what=f'three-launch Op D {case}')
        # the context head's patch-major bank (BankRef) instead of channel-major weights
        bank = HF.bank_pack(wt.to(dev), 0, m.hyper_params)
        yb = m(stage, HF.BankRef(bank, wt.shape[0], m.hyper_params, wt.shape[-2:]))
        assert torch.equal(yb, y_fused)


@pytest.mark.parametrize('case', [dict(skip=96, prev=0, cout=96, patch=1, grid=(16, 16), batch=4),     # HyperSeg-L level 0 (a bs-4 shard)
                                  dict(skip=34, prev=96, cout=34, patch=2, grid=(16, 16), batch=4),    # level 1
                                  dict(skip=10, prev=7, cout=40, patch=2, grid=(17, 33), batch=2),     # odd grid, cin = 19: odd -> the LDS-staged kernel
                                  dict(skip=20, prev=0, cout=5, patch=1, grid=(32, 35), batch=1),
                                  dict(skip=6, prev=4, cout=96, patch=1, grid=(40, 30), batch=1)])
def test_batched_tiny_patches_on_the_matrix_cores(HF, O, dev, case):
    """Op A, k = 1, >= 1024 patches of 1 or 4 pixels: the weight-stream kernel (hs_patch_conv_k1m.hip) behind hs_patch_conv_fwd,
    with and without a previous level, BatchNorm + ReLU epilogue, against the oracle's patch_conv_k1 on the stage input."""
    g = torch.Generator().manual_seed(case['skip'] * 7 + case['cout'])
    fh, fw = case['grid']
    h, w = fh * case['patch'], fw * case['patch']
    bsz = case['batch']
    skip = torch.randn(bsz, case['skip'], h, w, generator=g)
    prev = torch.randn(bsz, case['prev'], h // 2, w // 2, generator=g) if case['prev'] else None
    if prev is not None and (h % 2 or w % 2):
        pytest.skip('previous level needs an even map')
    cin = 2 + case['skip'] + case['prev']
    wt = torch.randn(bsz, case['cout'] * cin, fh, fw, generator=g) * (1.0 / cin) ** 0.5
    scale, shift = torch.rand(case['cout'], generator=g) + 0.5, torch.randn(case['cout'], generator=g) * 0.1
    ref = torch.relu(O.patch_conv_k1(O.stage_input(skip, prev), wt, case['cout']) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    stage = HF.StageInput(skip.to(dev), prev.to(dev) if prev is not None else None, coords=True)
    bank = HF.bank_pack(wt.to(dev), 0, wt.shape[1])
    y = HF.patch_conv(stage, (fh, fw), bank, case['cout'], scale=scale.to(dev), shift=shift.to(dev), act=HF.ACT_RELU)
    cmp(y, ref, what=f'batched tiny patches {case}')
    y0 = HF.patch_conv(stage, (fh, fw), bank, case['cout'])
    cmp(y0, O.patch_conv_k1(O.stage_input(skip, prev), wt, case['cout']), what=f'batched tiny patches, no epilogue {case}')


def test_fused_op_d_falls_back_when_regions_do_not_tile(HF, O, dev):
    """12 x 20 pixels at patch 4 do not tile into 8x8 regions, and 56 input channels are beyond the two-launch form: the module
    silently takes the three-launch route."""
    case = dict(skip=20, prev=34, cout=12, patch=4, grid=(3, 5))
    m, skip, prev, wt, ref = _op_d_case(case, dev, O)
    with torch.no_grad():
        stage = HF.StageInput(skip.to(dev), prev.to(dev), coords=True)
        assert m._forward_fused(stage, wt.to(dev)) is None
        cmp(m(stage, wt.to(dev)), ref, what='fallback Op D')


def test_full_config_l_bs32_properties(O, dev):
    """BASELINE config 4 at its full batch (HyperSeg-L 512x512, bs 32): frames are independent in eval mode, so the
    bs-32 decoder output must equal, frame by frame and bit for bit, the outputs of the per-GPU shards (bs 4, the
    configuration test_full_config[L] pins to the reference), whatever the position of a frame inside its batch."""
    d = build_decoder('L', O).to(dev)
    x, w = O.synth_decoder_inputs('L', batch=32, seed=0)
    x = [t.to(dev) for t in x]
    w = [t.to(dev) for t in w]
    with torch.no_grad():
        y = d(x, w)
        assert tuple(y.shape) == (32, 21, 512, 512) and bool(torch.isfinite(y).all())
        for lo in (0, 12, 28):
            ys = d([t[lo:lo + 4].contiguous() for t in x], [t[lo:lo + 4].contiguous() for t in w])
            assert torch.equal(ys, y[lo:lo + 4]), f'shard starting at frame {lo}'
    # the first shard is the fixture configuration of test_full_config[L] (same seed => same first frames? no: the
    # generator is consumed per tensor, so only the oracle is compared here, on one frame)
    ref = O.decoder_v0_1(O.config_plan('L'), O.synth_decoder_params(O.config_plan('L'), seed=0),
                         [t[5:6].cpu() for t in x], [t[5:6].cpu() for t in w])
    cmp(y[5:6], ref, what='L bs32 frame 5 vs oracle')


# ------------------------------------------------------------------------------ split arithmetic: range and scaling
def _fold(bn, eps=1e-5):
    sc = bn['weight'] / torch.sqrt(bn['running_var'] + eps)
    return sc, bn['bias'] - bn['running_mean'] * sc


def _op_c_case(O, cin_parts, cout, hid, patch, grid, seed, in_gain=1.0, w1_gain=1.0, w3_gain=1.0, w3_late_gain=1.0):
    """A v1_0 inverted residual at a decoder level shape with power-of-two gains on the inputs / weights (the BN that
    follows is rescaled so that the block computes the same function: ReLU6 stays exercised on both sides)."""
    skip_c, prev_c = cin_parts
    cin = 2 + skip_c + prev_c
    fh, fw = grid
    ph, pw = patch if isinstance(patch, tuple) else (patch, patch)
    h, w = fh * ph, fw * pw
    g = torch.Generator().manual_seed(seed)
    skip = torch.randn(1, skip_c, h, w, generator=g) * in_gain
    prev = torch.randn(1, prev_c, h // 2, w // 2, generator=g) * in_gain
    wt = torch.randn(1, cin * hid + 9 * hid + hid * cout, fh, fw, generator=g)
    wt[:, :cin * hid] *= (2.0 / cin) ** 0.5 * w1_gain
    wt[:, :cin * hid].view(1, hid, cin, fh, fw)[:, :, :2] *= in_gain      # the coordinates do not carry the input gain
    wt[:, cin * hid:cin * hid + 9 * hid] *= (2.0 / 9) ** 0.5
    w3 = wt[:, cin * hid + 9 * hid:].view(1, cout, hid, fh, fw)
    w3 *= (1.0 / hid) ** 0.5 * w3_gain
    w3[:, :, 32:] *= w3_late_gain                      # hidden chunks 2.. : the rows' running exponents must grow
    bns = []
    for n, gain in ((hid, in_gain * w1_gain), (hid, 1.0), (cout, 1.0)):
        bns.append({'weight': torch.rand(n, generator=g) + 0.5, 'bias': torch.randn(n, generator=g) * 0.1,
                    'running_mean': torch.randn(n, generator=g) * 0.1 * gain,
                    'running_var': (torch.rand(n, generator=g) * 1.5 + 0.5) * gain * gain})
    ref = O.patch_inverted_residual_v1(O.stage_input(skip, prev), wt, hid, cout, *bns)
    return skip, prev, wt, bns, ref


@pytest.mark.parametrize('gains', [dict(), dict(in_gain=2.0 ** 12, w1_gain=2.0 ** -9), dict(in_gain=2.0 ** -20, w1_gain=2.0 ** 14),
                                   dict(w3_gain=2.0 ** -12, w3_late_gain=2.0 ** 9), dict(in_gain=2.0 ** 16, w3_gain=2.0 ** 20)])
@pytest.mark.parametrize('shape', [((16, 16), 19, 78, 16, (2, 3)), ((6, 16), 16, 48, 8, (3, 2))])
def test_split_ir_ranges(HF, O, dev, ir_math, shape, gains):
    """The f16 split products carry power-of-two scales chosen from the data (per weight row, per position tile, a running
    exponent per pw3 row): inputs at 2^12 / 2^-20 / 2^30, weights at 2^-9 .. 2^20, and late hidden chunks 512x larger
    than the first (accumulator rescale) must all come out at f32 accuracy -- and identically in exact-f32 mode.
    (What the split form cannot carry is a dynamic range beyond ~2^18 INSIDE one reduction -- a weight 2^-20 of its row's
    maximum meeting an input 2^20 above its tile's: f16's exponent range; include/hyperseg_hip.h, hs_ir_math.)"""
    cin_parts, cout, hid, patch, grid = shape
    skip, prev, wt, bns, ref = _op_c_case(O, cin_parts, cout, hid, patch, grid, seed=11, **gains)
    stage = HF.StageInput(skip.to(dev), prev.to(dev), coords=True)
    bank = HF.bank_pack(wt.to(dev), 0, wt.shape[1])
    y = HF.patch_ir(stage, grid, bank, hid, cout, *[tuple(t.to(dev) for t in _fold(b)) for b in bns])
    assert bool(torch.isfinite(y).all())
    cmp(y, ref, what=f'Op C {shape} gains {gains} ({ir_math})')


# (skip, prev) channels, classes, hidden, patch (rows, cols), grid -- none of them a BASELINE level shape
GENERIC_OP_C = [((3, 16), 20, 42, (32, 32), (2, 2)),      # CamVid-L's 6th level (configs/train/camvid_efficientnet_b1_hyperseg-l.py:35-38), 20 classes
                ((5, 7), 9, 30, (16, 16), (2, 3)),        # odd channel counts everywhere: cin = 14 .. padded K slots, odd bank columns
                ((4, 8), 8, 28, (8, 16), (3, 2)),         # 8-row patches: the 16 x 8 region form
                ((16, 8), 19, 52, (16, 32), (2, 2)),      # HyperSeg-S level 4's channels on 16 x 32 patches (two regions per patch, side by side)
                ((6, 16), 16, 48, (16, 16), (3, 3)),      # level-3 channels on 16 x 16 patches
                ((16, 16), 32, 96, (16, 16), (2, 2))]     # the largest shape the kernel takes: 16 + 16 -> 96 -> 32


@pytest.mark.parametrize('shape', GENERIC_OP_C)
def test_op_c_generic_shapes_on_the_matrix_cores(HF, O, dev, shape):
    """dispatch is by RANGES (c_skip <= 16, c_prev <= 16, c_out <= 32, hid <= 96, patches >= 8 x 16), not by a
    table of BASELINE triples -- any such level gets the f16-split matrix-core kernel (hs_patch_irc.hip), says so through
    hs_patch_ir_route, and matches the oracle at the same tolerance."""
    cin_parts, cout, hid, patch, grid = shape
    skip, prev, wt, bns, ref = _op_c_case(O, cin_parts, cout, hid, patch, grid, seed=23)
    b, _, h, w = skip.shape
    assert HF.patch_ir_route((b, h, w), cin_parts[0], cin_parts[1], grid, hid, cout, math='auto') == 'split_mfma'
    stage = HF.StageInput(skip.to(dev), prev.to(dev), coords=True)
    bank = HF.bank_pack(wt.to(dev), 0, wt.shape[1])
    bnf = [tuple(t.to(dev) for t in _fold(bb)) for bb in bns]
    y = HF.patch_ir(stage, grid, bank, hid, cout, *bnf, math='split')
    cmp(y, ref, what=f'Op C generic {shape} (split)')
    y32 = HF.patch_ir(stage, grid, bank, hid, cout, *bnf, math='f32')           # generic vector-ALU kernel for most of these
    cmp(y32, ref, what=f'Op C generic {shape} (f32)')


def test_misaligned_bank_takes_the_direct_kernels(HF, O, dev):
    """the split kernel's LDS-DMA moves 16-byte pieces of the bank; a bank VIEW whose storage offset is not a
    multiple of 16 bytes (rows still contiguous, ld % 4 == 0) must fall back to the exact kernels -- same numbers -- instead
    of issuing misaligned 16-byte requests."""
    cin_parts, cout, hid, patch, grid = ((6, 16), 16, 48, (16, 16), (3, 3))
    skip, prev, wt, bns, ref = _op_c_case(O, cin_parts, cout, hid, patch, grid, seed=29)
    stage = HF.StageInput(skip.to(dev), prev.to(dev), coords=True)
    aligned = HF.bank_pack(wt.to(dev), 0, wt.shape[1])
    buf = torch.empty(aligned.numel() + 4, device=dev)
    for off in (1, 2, 3):
        bank = buf[off:off + aligned.numel()].view_as(aligned)
        bank.copy_(aligned)
        assert bank.data_ptr() % 16 == 4 * off
        bnf = [tuple(t.to(dev) for t in _fold(bb)) for bb in bns]
        y = HF.patch_ir(stage, grid, bank, hid, cout, *bnf, math='split')
        cmp(y, ref, what=f'Op C, bank offset {4 * off} bytes (split requested)')


def test_split_and_exact_modes_agree_on_flips(HF, O, dev):
    """HyperSeg-M at 1024x512: the two arithmetic modes give the same mask except where the oracle's own top-2 margin is
    below MARGIN, and logits within REL_TOL of each other."""
    d = build_decoder('M', O).to(dev)
    x, s = O.synth_decoder_inputs('M', batch=1, seed=3)
    x = [t.to(dev) for t in x]
    ys = {}
    with torch.no_grad():
        for mode in ('split', 'f32'):
            prev = HF.set_ir_math(mode)
            try:
                ys[mode] = d(x, s.to(dev))
            finally:
                HF.set_ir_math(prev)
    assert HF.get_ir_math() == HF.DEFAULT_IR_MATH == 'f32'  # no override left behind; the modules' default is the exact form
    e = float((ys['split'] - ys['f32']).abs().max() / ys['f32'].abs().max())
    assert e < REL_TOL, e
    top2 = ys['f32'].topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    flips = ys['split'].argmax(1) != ys['f32'].argmax(1)
    assert int((flips & (margin > MARGIN)).sum()) == 0


# ------------------------------------------------------------------------------ bank generated inside the consumer
GEN_CASES = [
    # skip, prev, cout, patch, grid, batch, signal channels, groups  (rows padded to a multiple of groups like the reference)
    dict(skip=80, prev=0, cout=64, patch=1, grid=(4, 8), batch=1, cs=416, groups=32),      # HyperSeg-M level 0: 2 channels / group
    dict(skip=28, prev=64, cout=32, patch=2, grid=(4, 8), batch=1, cs=224, groups=16),     # level 1, bilinear previous level
    dict(skip=10, prev=32, cout=16, patch=4, grid=(4, 8), batch=1, cs=128, groups=8),      # level 2
    dict(skip=10, prev=32, cout=16, patch=4, grid=(3, 5), batch=2, cs=256, groups=32),     # CamVid-S level 2: a channel spans
                                                                                          # two groups; 30 patches (ragged tile)
    dict(skip=128, prev=0, cout=32, patch=1, grid=(3, 4), batch=1, cs=576, groups=32),     # HyperSeg-S level 0: K = 18
    dict(skip=5, prev=0, cout=7, patch=8, grid=(2, 2), batch=1, cs=24, groups=3),          # odd everything, 64-pixel patches
]


@pytest.mark.parametrize('case', GEN_CASES)
def test_bank_in_consumer_vs_oracle(HF, O, dev, case):
    """hs_patch_conv_gen_fwd (signal2weights + k=1 patch conv + BN + ReLU, the bank never in HBM) == the oracle's
    signal2weights -> patch_conv_k1 -> bn -> relu, and == the two-launch HIP route."""
    c = case
    g = torch.Generator().manual_seed(3 + c['cs'])
    fh, fw = c['grid']
    h, w = fh * c['patch'], fw * c['patch']
    cin = 2 + c['skip'] + c['prev']
    hp = c['cout'] * cin
    wc = -(-hp // c['groups']) * c['groups']
    sidx = 8
    s = torch.relu(torch.randn(c['batch'], sidx + c['cs'] + 5, fh, fw, generator=g))
    wsw = torch.randn(wc, c['cs'] // c['groups'], 1, 1, generator=g) * (c['groups'] / c['cs']) ** 0.5
    skip = torch.randn(c['batch'], c['skip'], h, w, generator=g)
    prev = torch.randn(c['batch'], c['prev'], h // 2, w // 2, generator=g) if c['prev'] else None
    bn = dict(weight=torch.rand(c['cout'], generator=g) + 0.5, bias=torch.randn(c['cout'], generator=g) * 0.1,
              running_mean=torch.randn(c['cout'], generator=g) * 0.1, running_var=torch.rand(c['cout'], generator=g) + 0.5)
    wt = O.signal2weights(s, wsw, sidx, c['cs'], c['groups'], hp)
    ref = O.act(O.bn_eval(O.patch_conv_k1(O.stage_input(skip, prev), wt, c['cout']), bn), O.ACT_RELU)
    with torch.no_grad():
        sd, stage = s.to(dev), HF.StageInput(skip.to(dev), prev.to(dev) if prev is not None else None, coords=True)
        layer = dict(wsw_t=wsw.reshape(wc, -1).t().contiguous().to(dev), signal_index=sidx, signal_channels=c['cs'],
                     groups=c['groups'], rows=hp)
        scale, shift = HF.bn_fold(*(bn[k].to(dev) for k in ('weight', 'bias', 'running_mean', 'running_var')))
        y = HF.patch_conv_gen(stage, HF.SignalRef(sd, layer), c['cout'], scale, shift, HF.ACT_RELU)
        assert y is not None
        cmp(y, ref, what=f'bank-in-consumer {c}')
        bank = HF.signal2weights_multi(sd, [layer])[0].bank
        y2 = HF.patch_conv(stage, (fh, fw), bank, c['cout'], 1, 0, 'zeros', 1, scale, shift, HF.ACT_RELU)
        cmp(y, y2.cpu(), what='vs the two-launch route')


def test_bank_in_consumer_is_what_the_decoder_runs(HF, O, dev, monkeypatch):
    """With the (opt-in) fusion on, the HyperSeg-M decoder's levels 0-2 go through hs_patch_conv_gen_fwd and only levels
    3-4 get a materialised bank; with it off (the default) the output is the same to rounding."""
    d = build_decoder('M', O).to(dev)
    x, s = O.synth_decoder_inputs('M', batch=1, seed=0, size=(128, 256))
    x, s = [t.to(dev) for t in x], s.to(dev)
    calls = {'gen': 0, 'layers': []}
    gen, multi = HF.patch_conv_gen, HF.signal2weights_multi
    monkeypatch.setattr(HF, 'patch_conv_gen', lambda *a, **k: (calls.__setitem__('gen', calls['gen'] + 1), gen(*a, **k))[1])
    monkeypatch.setattr(HF, 'signal2weights_multi', lambda sig, layers: (calls['layers'].append(len(layers)), multi(sig, layers))[1])
    monkeypatch.setattr(HF, 'BANK_IN_CONSUMER_MAX_PIXELS', 64)
    with torch.no_grad():
        y = d(x, s)
        assert calls['gen'] == 3 and calls['layers'] == [2]
        monkeypatch.setattr(HF, 'BANK_IN_CONSUMER_MAX_PIXELS', 0)
        y0 = d(x, s)
        assert calls['gen'] == 3 and calls['layers'] == [2, 5]
    cmp(y, y0.cpu(), what='fused vs materialised banks')


@pytest.mark.parametrize('chain', [False, True], ids=['three_launches', 'chain_k1'])
def test_two_python_threads_two_streams_through_the_decoder(O, HF, dev, chain):
    """SURVEY 8b "Threading" / VERDICT r4 weak #4: the reference's multi-GPU mode is nn.DataParallel, i.e. ONE Python thread per replica
    (parallel_apply; hyperseg/train.py:242-243, test_fps.py:155-156) through module objects whose non-tensor attributes -- this
    package's host-side caches -- are SHARED by reference between the replicas.  Here: two threads, each on its own HIP stream, push
    different frames through the same MultiScaleDecoder 40 times in different math modes (thread-local ir_math_scope) while a third
    thread keeps invalidating every cache (bump_weights_epoch: what a BatchNorm update or a graph replay does), so folded BatchNorm
    affines, transposed and packed signal2weights weights are rebuilt concurrently all the time.  Every output must equal the serial
    result bit for bit; no exception ("dictionary changed size during iteration" in round 4's functional._S2W_BLK sweep).
    Three phases, so that a failure says WHAT breaks: (A) one thread on a side stream, (B) two threads, (C) two threads + the invalidator."""
    import threading
    import time
    d = build_decoder('M', O).to(dev).eval()
    # chain_k1 (VERDICT r5 #4 / ADVICE r5): levels 0-2 as hs_k1_chain_fwd, whose workspace carries protocol state between launches and
    # whose grid must be resident at once -- one workspace per stream (functional.K1Chain) and chained launches ordered across streams
    # (functional.ChainGate), so the same bit-equality holds and the kernel's error word stays 0
    d.chain_k1 = chain
    frames = [O.synth_decoder_inputs('M', batch=1, seed=k, size=(128, 256)) for k in (0, 1)]
    frames = [([t.to(dev) for t in x], s.to(dev)) for x, s in frames]
    modes = ['f32', 'split']
    ref = []
    with torch.no_grad():
        for (x, s), mode in zip(frames, modes):
            with HF.ir_math_scope(mode):
                ref.append(d(x, s).clone())
                again = d(x, s)
            assert torch.equal(again, ref[-1]), f'the serial forward itself is not reproducible in mode {mode}'
    assert not torch.equal(ref[0], ref[1])
    torch.cuda.synchronize()

    def phase(which, invalidate, n_iter=40):
        outs, errors = {i: [] for i in which}, []
        start = threading.Barrier(len(which) + (1 if invalidate else 0))
        done = threading.Event()

        def replica(i):
            try:
                x, s = frames[i]
                stream = torch.cuda.Stream(dev)
                start.wait()
                with torch.no_grad(), torch.cuda.stream(stream), HF.ir_math_scope(modes[i]):
                    for _ in range(n_iter):
                        outs[i].append(d(x, s))
                stream.synchronize()
            except BaseException as e:          # noqa: BLE001 -- re-raised on the main thread
                errors.append(e)

        def invalidator():
            start.wait()
            while not done.is_set():
                HF.bump_weights_epoch()
                time.sleep(0.0003)
        threads = [threading.Thread(target=replica, args=(i,)) for i in which]
        inv = threading.Thread(target=invalidator) if invalidate else None
        for t in threads + ([inv] if inv else []):
            t.start()
        for t in threads:
            t.join(120)
        done.set()
        if inv:
            inv.join(10)
        assert not errors, errors
        torch.cuda.synchronize()
        report = []
        for i in which:
            assert len(outs[i]) == n_iter
            bad = [k for k, y in enumerate(outs[i]) if not torch.equal(y, ref[i])]
            if bad:
                worst = max(float((outs[i][k].double() - ref[i].double()).abs().max()) for k in bad)
                nan = sum(int(torch.isnan(outs[i][k]).sum()) for k in bad)
                report.append(f'thread {i} ({modes[i]}): {len(bad)} of {n_iter} outputs differ from the serial result (first {bad[:6]}), '
                              f'max |diff| {worst:.3e} of scale {float(ref[i].abs().max()):.3e}, {nan} NaNs')
        return report
    for name, which, inv in (('A: one thread, side stream', (0,), False), ('A2: one thread, side stream, split', (1,), False),
                             ('B: two threads', (0, 1), False), ('C: two threads + cache invalidator', (0, 1), True)):
        rep = phase(which, inv)
        assert not rep, f'phase {name}: ' + '; '.join(rep)
    assert getattr(HF._ir_math_local, 'mode', None) is None          # the scopes were the threads' own: nothing leaked into this one
    if chain:
        kc = d._k1_chain
        assert kc is not None and kc._ws, 'the chain refused a shape it is built for'
        assert len(kc._taken) >= 3, 'main stream + one workspace per replica stream'
        assert kc.error_word() == 0
        kc.request_error_copy(dev)
        torch.cuda.synchronize()
        kc.check_errors()                                             # pinned mirrors: nothing abandoned a wait
        # ... and a raised error word is what check_errors reports: poke one workspace, mirror it, expect the refusal of the next frame
        ws = next(iter(kc._taken.values()))[0]
        ws[:1].view(torch.int32)[:1].fill_(0x51)
        kc.request_error_copy(dev)
        torch.cuda.synchronize()
        with pytest.raises(RuntimeError, match='abandoned a wait'):
            with torch.no_grad():
                d(*frames[0])
        kc.reset()
        with torch.no_grad(), HF.ir_math_scope(modes[0]):
            assert torch.equal(d(*frames[0]), ref[0])


@pytest.mark.parametrize('name,size,with_ir', [('M', None, False), ('M', (128, 256), False), ('M', (32, 64), False), ('Sc', None, False),
                                               ('S', None, False),          # the unify decoder, 1152 cells: five 32.5 KB workgroups per CU
                                               ('M', None, True), ('M', (128, 256), True), ('M', (32, 64), True), ('Sc', None, True)])
def test_k1_chain_equals_the_three_launches(O, HF, dev, name, size, with_ir):
    """hs_k1_chain_fwd (levels 0-2 as ONE launch with in-launch neighbour hand-offs, csrc/hs_k1_chain.hip) against the three
    hs_patch_conv_fwd launches it replaces, through the whole decoder: full HyperSeg-M (512 cells: the whole grid resident) and CamVid-S
    grids, a 4 x 8 grid and a 1 x 2 grid (every cell on the border: all hand-offs clamped).  Same operations per output, another summation
    order: rounding-level agreement.  Six calls in a row (the generation counter in the workspace advances per call), then the same
    decoder captured in a HIP graph and replayed (kernel arguments frozen: the generation must come from memory), and the error word
    (a workgroup that gave up waiting for a neighbour) must stay 0.  ``with_ir``: the first inverted-residual level (8 x 8-pixel
    patches) rides in the same launch (hs_decoder_chain_fwd) -- exact f32 on the matrix cores, against hs_patch_ir_fwd in every math mode
    the suite runs (the reference tolerance is the same for all of them)."""
    d = build_decoder(name, O).to(dev).eval()
    d.chain_ir = with_ir
    kw = {} if size is None else dict(size=size)
    frames = [O.synth_decoder_inputs(name, batch=1, seed=k, **kw) for k in (0, 1)]
    frames = [([t.to(dev) for t in x], s.to(dev)) for x, s in frames]
    with torch.no_grad():
        ref = [d(x, s).clone() for x, s in frames]
        d.chain_k1 = True
        for it in range(6):
            x, s = frames[it & 1]
            y = d(x, s)
            assert d._k1_chain is not None and d._k1_chain._ws, 'the chain refused a shape it is built for'
            assert any((k[-1] is not None) == with_ir for k in d._k1_chain._ws), 'the inverted residual did not ride in the chain'
            assert rel_err(y.cpu(), ref[it & 1].cpu()) < REL_TOL, f'call {it}'
        assert d._k1_chain.error_word() == 0
        # graph replay: same launch parameters every time
        x, s = frames[0]
        xs, ss = [t.clone() for t in x], s.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            d(xs, ss)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            yg = d(xs, ss)
        for it in range(5):
            for dst, src in zip(xs + [ss], frames[it & 1][0] + [frames[it & 1][1]]):
                dst.copy_(src)
            g.replay()
            torch.cuda.synchronize()
            assert rel_err(yg.cpu(), ref[it & 1].cpu()) < REL_TOL, f'replay {it}'
        assert d._k1_chain.error_word() == 0


@pytest.mark.parametrize('case', [dict(b=1, fh=4, fw=8, hps=(35, 1000, 147 * 3 + 1), grps=(4, 16, 3), cs=(16, 64, 24)),          # rows per group 9, 63, 148: every alignment
                                  dict(b=3, fh=4, fw=6, hps=(9, 70, 257), grps=(1, 2, 1), cs=(8, 12, 5)),                        # 72 patches: a partial patch block; 257 rows: a 1-row block
                                  dict(b=1, fh=16, fw=32, hps=(2352, 4216), grps=(16, 4), cs=(192, 320))])                       # HyperSeg-M levels 3-4 (147, 1054 rows per group)
def test_signal2weights_blocked_stores_stay_inside_their_rows(HF, O, dev, case):
    """Round 6: the blocked signal2weights writes 16-byte quads of the bank rows, single floats only at the two ends of a block whose first
    row is not a multiple of 4.  Into a buffer pre-filled with a sentinel: every bank value equals the oracle's, and NOTHING else is touched --
    the pad columns [hp, ld) of every row and the floats behind the last bank keep the sentinel (an over-wide store at a block edge would
    show there or in the neighbouring group's rows, which the value check covers)."""
    g = G(2001)
    c = case
    c_sig = max(c['cs']) + 5
    s = torch.randn(c['b'], c_sig, c['fh'], c['fw'], generator=g).clamp(min=-0.5)
    layers, refs = [], []
    for hp, grp, cs in zip(c['hps'], c['grps'], c['cs']):
        rows = O.next_multiply(hp, grp)
        w = torch.randn(rows, cs // grp, 1, 1, generator=g)
        layers.append(dict(wsw_t=w.reshape(rows, -1).t().contiguous().to(dev), signal_index=2, signal_channels=cs, groups=grp, rows=hp))
        refs.append(O.signal2weights(s, w, 2, cs, grp, hp).permute(0, 2, 3, 1).reshape(-1, hp))
    n = HF.bank_floats(s, layers)
    sentinel = -12345.5
    buf = torch.full((n + 64,), sentinel, device=dev)
    out = HF.signal2weights_multi(s.to(dev), layers, buf=buf[:n])
    torch.cuda.synchronize()
    for ref, o, hp in zip(refs, out, c['hps']):
        cmp(o.bank[:, :hp], ref, what=f's2w rows {hp}')
        if o.bank.shape[1] > hp:
            assert bool((o.bank[:, hp:] == sentinel).all()), f'pad columns of the {hp}-row bank were written'
    assert bool((buf[n:] == sentinel).all()), 'floats behind the last bank were written'
