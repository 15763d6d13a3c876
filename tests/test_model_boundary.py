"""Drop-in boundary of the whole model: factory kwargs / arch strings, state-dict keys
and shapes, encoder + context head numerics (stock PyTorch, CPU) and -- on the GPU -- the end-to-end logits
of HyperGen against the reference's (fixture model_M.npz, name-keyed weights)."""
import hashlib

import pytest
import torch

from conftest import G, rel_err
from hyperseg_amd.utils.synthetic import fill_by_name


def _h(s):
    return int.from_bytes(hashlib.sha256(s.encode()).digest()[:7], 'little')


@pytest.fixture(scope='module')
def model_m():
    from hyperseg_amd import configs
    return fill_by_name(configs.build('hyperseg-m').eval(), seed=11)


MODELS = {'M': 'hyperseg-m', 'S': 'hyperseg-s', 'L': 'hyperseg-l', 'Lc': 'hyperseg-l-camvid'}


@pytest.mark.parametrize('tag', ['S', 'L', 'Lc'])
def test_state_dict_and_context_head_other_variants(golden, tag):
    """unify (HyperSeg-S), v0_1 (HyperSeg-L) and the six-level v1_0 model (CamVid HyperSeg-L, round 6): key/shape hashes +
    encoder/context-head numerics on CPU."""
    from hyperseg_amd import configs
    g = golden(f'model_{tag}')
    m = fill_by_name(configs.build(MODELS[tag]).eval(), seed=11)
    sd = m.state_dict()
    keys = [k for k in sd if 'num_batches' not in k]
    assert len(keys) == int(g['n_keys'])
    assert _h(' '.join(keys)) == int(g['key_hash'][0])
    assert _h(' '.join(str(tuple(sd[k].shape)) for k in keys)) == int(g['shape_hash'][0])
    with torch.no_grad():
        feats = m.backbone(g['x'])
        sig = m.weight_mapper(feats[-1])
    if isinstance(sig, (list, tuple)):          # v0_1: the "signal" is the list of per-level weight tensors
        sig = torch.cat([t[:, ::97] for t in sig], dim=1)
    assert rel_err(sig[:, ::13], g['signal']) < 1e-4
    if tag == 'S':
        assert [(w.signal_index, w.signal_channels) for w in m.decoder.weight_blocks] == \
               [(0, 576), (576, 128), (704, 64), (768, 512)]
        assert tuple(sd['decoder.weight_blocks.3.signal2weights.weight'].shape) == (3680, 32, 1, 1)
    elif tag == 'L':
        assert tuple(sd['weight_mapper.out_conv.conv_1.weight'].shape) == (4496, 17, 1, 1)
        assert m.decoder.param_groups == [9408, 4488, 6624, 1716, 992, 902]
    else:                                       # CamVid-L: six levels, hyper-parameter counts as the reference builds them
        assert m.decoder.levels == 6 and m.decoder.param_groups == [5248, 3008, 704, 2352, 2068, 1764]
        assert tuple(sd['decoder.level_5.0.signal2weights.weight'].shape) == (1768, 16, 1, 1)      # next_multiply(1764, 8 groups)


def test_state_dict_contract(golden, model_m):
    g = golden('model_M')
    sd = model_m.state_dict()
    keys = [k for k in sd if 'num_batches' not in k]
    assert len(keys) == int(g['n_keys'])
    assert _h(' '.join(keys)) == int(g['key_hash'][0]), 'state-dict key names differ from the reference'
    assert _h(' '.join(str(tuple(sd[k].shape)) for k in keys)) == int(g['shape_hash'][0])
    # Appendix C spot checks
    assert tuple(sd['decoder.level_0.0.0.signal2weights.weight'].shape) == (5248, 13, 1, 1)
    assert tuple(sd['decoder.level_4.0.signal2weights.weight'].shape) == (4216, 80, 1, 1)
    assert tuple(sd['decoder.coord512_1024'].shape) == (1, 2, 512, 1024)
    assert model_m.backbone.feat_channels == [16, 6, 10, 28, 80, 1280]


def test_encoder_and_context_head_cpu(golden, model_m):
    g = golden('model_M')
    with torch.no_grad():
        feats = model_m.backbone(g['x'])
        sig = model_m.weight_mapper(feats[-1])
    assert [float(f.abs().max()) for f in feats] == pytest.approx(list(g['feat_absmax'].tolist()), rel=1e-3)
    assert rel_err(sig[:, ::13], g['signal']) < 1e-4


def test_obj_factory_contract():
    from functools import partial
    from hyperseg_amd.utils.obj_factory import obj_factory, partial_obj_factory
    conv = obj_factory('hyperseg.models.layers.meta_conv.MetaConv2d(3, 6, kernel_size=3)', padding=1)
    assert conv.hyper_params == 6 * 3 * 9 and conv.padding == (1, 1)
    assert isinstance(obj_factory('nn.ReLU(True)'), torch.nn.ReLU)
    p = partial_obj_factory('hyperseg.models.layers.meta_patch.MetaPatchConv2d(4, 8)', kernel_size=1)
    assert isinstance(p, partial) and p().hyper_params == 32
    assert obj_factory([partial(int, '7'), 3]) == [7, 3]
    wg = [32, 16, 8, 16, 4]
    from hyperseg_amd.models.hyperseg_v1_0 import hyperseg_efficientnet
    hyperseg_efficientnet('efficientnet-b1', levels=2, out_feat_scale=[1., .25, .25, .25, .25],
                          kernel_sizes=[1, 1, 1, 3, 3], level_channels=[64, 32, 16, 16, 16], expand_ratio=2,
                          weight_groups=wg, num_classes=19)
    assert wg == [32, 16, 8, 16, 4], 'the caller\'s weight_groups list must survive construction (Appendix D-3)'


def test_decoder_refuses_cpu(model_m):
    from hyperseg_amd._hip import HipLibraryError
    with torch.no_grad(), pytest.raises(HipLibraryError):
        model_m(torch.rand(1, 3, 64, 64, generator=G(1027)))


def test_graphed_model_host_logic(model_m):
    """utils.inference.GraphedModel off the GPU: nothing on a CPU model is graphable, so every call is handed to the wrapped
    model unchanged (which, without a GPU, fails loudly in the decoder -- no CPU fallback); the harness hook, the graph
    table and reset() behave; state-dict keys are the wrapped model's under ``model.``."""
    from hyperseg_amd._hip import HipLibraryError
    from hyperseg_amd.fps import measure_fps, synthetic_batches
    from hyperseg_amd.utils.inference import GraphedModel
    served = GraphedModel(model_m, clone_output=True)
    x = torch.rand(1, 3, 64, 64, generator=G(1028))
    assert served.accepts_host_input and not served._graphable(x) and not served._graphable([x, x])
    with torch.no_grad(), pytest.raises(HipLibraryError):
        served(x)
    assert served._graphs == {}
    assert set(served.state_dict()) == {'model.' + k for k in model_m.state_dict()}
    served._graphs['stale'] = None
    model_m.load_state_dict(model_m.state_dict())               # the post-hook drops captured graphs
    assert served._graphs == {}
    # measure_fps hands a wrapper that accepts host input the (pinned) host batch itself
    seen = []

    class Probe(torch.nn.Module):
        accepts_host_input = True

        def forward(self, t):
            seen.append(t.device.type)
            return torch.zeros(t.shape[0], 3, *t.shape[-2:])
    res = measure_fps(Probe(), synthetic_batches(2, 1, (8, 8), 3, torch.device('cpu')), torch.device('cpu'), 3, passes=1)
    assert seen == ['cpu', 'cpu'] and res['frames'] == 2


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['S', 'L', 'Lc'])
def test_model_end_to_end_other_variants(golden, tag):
    from hyperseg_amd import configs
    g = golden(f'model_{tag}')
    dev = torch.device('cuda:0')
    m = fill_by_name(configs.build(MODELS[tag]).eval(), seed=11).to(dev)
    with torch.no_grad():
        y = m(g['x'].to(dev)).cpu()
    assert list(y.shape) == [int(v) for v in g['y_shape']]
    ys = y[:, :, 1::5, 2::7]
    assert float((ys - g['y']).abs().max()) < 1e-3 * float(g['y_absmax'])
    ok = g['margin'] > 1e-3 * float(g['y_absmax'])
    assert bool((ys.argmax(1).to(torch.uint8)[ok] == g['mask'][ok]).all())


@pytest.mark.gpu
def test_model_m_end_to_end(golden, model_m):
    g = golden('model_M')
    dev = torch.device('cuda:0')
    m = model_m.to(dev)
    with torch.no_grad():
        y = m(g['x'].to(dev)).cpu()
    assert list(y.shape) == [int(v) for v in g['y_shape']]
    ys = y[:, :, 1::5, 2::7]
    # 25 stock conv layers in front of the decoder: MIOpen vs the CPU reference differ at the 1e-5 level
    assert float((ys - g['y']).abs().max()) < 1e-3 * float(g['y_absmax'])
    ok = g['margin'] > 1e-2 * float(g['y_absmax']) * 1e-1
    assert bool((ys.argmax(1).to(torch.uint8)[ok] == g['mask'][ok]).all())
    model_m.cpu()


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['M', 'S', 'L', 'Lc'])
def test_segment_equals_argmax_of_forward(golden, tag):
    """HyperGen.segment (argmax fused into the final upsample) == forward(x).argmax(1), bit for bit."""
    from hyperseg_amd import configs
    g = golden(f'model_{tag}')
    dev = torch.device('cuda:0')
    m = fill_by_name(configs.build(MODELS[tag]).eval(), seed=11).to(dev)
    x = g['x'].to(dev)
    with torch.no_grad():
        masks = m.segment(x)
        ref = m(x).argmax(1)
    assert masks.dtype == torch.uint8 and masks.shape == ref.shape
    assert bool((masks.long() == ref).all())


@pytest.mark.gpu
@pytest.mark.parametrize('prepared', [False, True])
def test_pyramid_hflip_inference_vs_reference(golden, prepared):
    """HyperGen's list-input mode (hyperseg_v1_0.py:70-91): two pyramid scales, horizontal-flip augmentation (max over the
    flip pair), mean over scales, coarse scale resized to the first -- against the reference's own output
    (fixture model_M_pyramid.npz), with the stock and with the prepared encoder."""
    from hyperseg_amd import configs
    from hyperseg_amd.utils.inference import prepare_for_inference
    g = golden('model_M_pyramid')
    dev = torch.device('cuda:0')
    m = fill_by_name(configs.build('hyperseg-m').eval(), seed=11)
    assert m.inference_hflip and m.inference_gather == 'mean'
    if prepared:
        prepare_for_inference(m, fold_bn=False, fused_depthwise=True)
    m = m.to(dev)
    with torch.no_grad():
        y = m([g['x0'].to(dev), g['x1'].to(dev)]).cpu()
    assert list(y.shape) == [int(v) for v in g['y_shape']]
    ys = y[:, :, 1::3, 2::5]
    assert float((ys - g['y']).abs().max()) < 1e-3 * float(g['y_absmax'])
    ok = g['margin'] > 1e-2 * float(g['y_absmax'])
    assert bool((ys.argmax(1).to(torch.uint8)[ok] == g['mask'][ok]).all())


@pytest.mark.gpu
def test_fps_harness_and_bn_removal():
    """hyperseg_amd.fps (the test_fps.py counterpart) end to end on the GPU, incl. the reference's BN -> identity switch:
    with every BatchNorm removed the fused decoder kernels must equal the same model with identity-valued BatchNorms."""
    import copy
    from hyperseg_amd import configs, fps
    dev = torch.device('cuda:0')
    model = fill_by_name(configs.build('hyperseg-m').eval(), seed=4)
    ident = copy.deepcopy(model)
    for m in ident.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            with torch.no_grad():
                m.weight.fill_(1.0); m.bias.zero_(); m.running_mean.zero_(); m.running_var.fill_(1.0 - m.eps)
    fps.remove_bn(model)
    assert not any(isinstance(m, torch.nn.BatchNorm2d) for m in model.modules())
    x = torch.rand(1, 3, 128, 256, generator=G(1029))
    with torch.no_grad():
        a, b = model.to(dev)(x.to(dev)).cpu(), ident.to(dev)(x.to(dev)).cpu()
    assert rel_err(a, b) < 1e-4
    batches = fps.synthetic_batches(3, 1, (128, 256), 19, dev)
    res = fps.measure_fps(model, batches, dev, 19)
    assert res['frames'] == 3 and res['fps'] > 1.0 and 0.0 <= res['mean_iou'] <= 1.0


@pytest.mark.gpu
@pytest.mark.parametrize('batch', [1, 2])
def test_inference_prep_matches_stock_encoder(batch):
    """prepare_for_inference (fused depthwise + BN + swish + SE kernels, gate folded into the project conv) leaves the
    encoder's outputs unchanged: features of the prepared model == features of the stock model (rel 1e-5)."""
    import copy
    from hyperseg_amd import configs
    from hyperseg_amd.utils.inference import prepare_for_inference
    dev = torch.device('cuda:0')
    stock = fill_by_name(configs.build('hyperseg-m').eval(), seed=5)
    fused = copy.deepcopy(stock)
    n_bn = len([m for m in stock.backbone.modules() if isinstance(m, torch.nn.BatchNorm2d)])
    prepare_for_inference(fused, fold_bn=False, fused_depthwise=True)
    assert all(b._fused_dw is not None for b in fused.backbone._blocks)
    assert len([m for m in fused.backbone.modules() if isinstance(m, torch.nn.BatchNorm2d)]) == n_bn
    assert list(fused.state_dict()) == list(stock.state_dict())          # checkpoints still round-trip
    stock, fused = stock.to(dev), fused.to(dev)
    x = torch.rand(batch, 3, 128, 192, generator=G(1030)).to(dev)
    with torch.no_grad():
        fs, ff = stock.backbone(x), fused.backbone(x)
        for a, b in zip(fs, ff):
            assert rel_err(b.cpu(), a.cpu()) < 2e-5
        assert rel_err(fused(x).cpu(), stock(x).cpu()) < 1e-4


@pytest.mark.parametrize('config,split', [('hyperseg-m', False), ('hyperseg-s', False), ('hyperseg-l', False),
                                          ('hyperseg-m', True), ('hyperseg-l', True)])
def test_deferred_bn_shift_algebra_cpu(monkeypatch, config, split):
    """Host logic of utils.inference (no GPU): the deferred-BN-shift bookkeeping of the fused MBConv blocks is exact.
    The HIP entry points are replaced by plain-torch stand-ins of their documented semantics, the fused encoder is
    walked by hand (the product path refuses CPU tensors) and must reproduce the stock encoder's features.
    ``split``: the opt-in hs_gemm_split_fwd route -- the REAL host-side weight preparation (row scaling, f16 pieces, fragment
    order) feeds a stand-in that rebuilds W from it, so both the wiring and the preparation are exercised."""
    import copy
    import torch.nn.functional as F
    from hyperseg_amd import configs, functional as HF
    from hyperseg_amd.utils.inference import prepare_for_inference

    def act_of(t, act):
        return t * torch.sigmoid(t) if act == 3 else (torch.relu(t) if act == 1 else (t.clamp(0, 6) if act == 2 else t))

    def dw(x, weight, stride, pad_top, pad_left, out_size, scale=None, shift=None, act=0, pool=False, in_scale=None,
           in_shift=None, se=None):                          # se: the stand-in leaves the gate to se_gate below (gated = False)
        if in_scale is not None:
            x = act_of(x * in_scale.view(1, -1, 1, 1) + in_shift.view(1, -1, 1, 1), 3)
        k = weight.shape[-1]
        ho, wo = out_size
        pb, pr = (ho - 1) * stride + k - x.shape[2] - pad_top, (wo - 1) * stride + k - x.shape[3] - pad_left
        y = F.conv2d(F.pad(x, (pad_left, pr, pad_top, pb)), weight, stride=stride, groups=x.shape[1])
        y = act_of(y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1), act)
        return (y, y.sum((2, 3)).view(-1, 1)) + ((False,) if se is not None else ())

    def se_gate(partial, batch, hw, w_reduce, b_reduce, w_expand_t, b_expand, w_proj=None, out_scale=None):
        pooled = partial.sum(1).view(batch, -1) / hw
        z = act_of(pooled @ w_reduce.flatten(1).t() + b_reduce, 3)
        gate = torch.sigmoid(z @ w_expand_t + b_expand)
        if w_proj is None:
            return gate
        ws = w_proj.flatten(1)[None] * gate[:, None, :]
        if out_scale is not None:
            ws = ws * out_scale[None, :, None]
        return ws[..., None, None]

    def pointwise(x, weight, gate=None, scale=None, shift=None, act=0, residual=None):
        if gate is not None:
            x = x * gate[:, :, None, None]
        y = act_of(F.conv2d(x, weight) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1), act)
        return y if residual is None else y + residual

    def affine(x, scale, shift, act=0, residual=None):
        y = x if scale is None else x * scale.view(1, -1, 1, 1)
        y = act_of(y + shift.view(1, -1, 1, 1), act)
        x.copy_(y if residual is None else y + residual)
        return x

    def expand_dw(x, w_expand, scale0, shift0, w_dw, stride, pad_top, pad_left, out_size, scale1, shift1, pool=True, se=None):
        hmid = pointwise(x, w_expand.view(w_expand.shape[0], -1, 1, 1), None, scale0, shift0, 3)
        return dw(hmid, w_dw, stride, pad_top, pad_left, out_size, scale1, shift1, act=3, pool=True, se=se)

    used = {'n': 0, 'gated': 0, 'accumulated': 0}

    def gemm_split(sw, x, gate=None, shift=None, act=0, residual=None, out=None):
        rt = sw.frag.shape[0]
        back = sw.frag.permute(2, 0, 4, 1, 3, 5).reshape(2, 16 * rt, sw.kp).float()        # [piece][row][k]
        w = ((back[0] + back[1]) * sw.inv[:, None])[:sw.c_out, :sw.c_in]
        y = F.conv2d(x if gate is None else x * gate[:, :, None, None], w[:, :, None, None])
        if shift is not None:
            y = y + shift.view(1, -1, 1, 1)
        y = act_of(y, act)
        if residual is not None:
            y = y + residual
        used['n'] += 1
        used['gated'] += gate is not None
        used['accumulated'] += residual is not None and residual is out
        if out is None:
            return y
        assert out.shape == y.shape
        return out.copy_(y)

    def weight_of(sw):
        rt = sw.frag.shape[0]
        back = sw.frag.permute(2, 0, 4, 1, 3, 5).reshape(2, 16 * rt, sw.kp).float()
        return ((back[0] + back[1]) * sw.inv[:, None])[:sw.c_out, :sw.c_in]

    def gemm_split_conv2x2(sw, x, shift=None, act=0, out=None, pool_partial=None):           # K = 4 Cin in the conv weight's flatten order
        y = F.conv2d(x, weight_of(sw).reshape(sw.c_out, x.shape[1], 2, 2), stride=2)
        y = act_of(y if shift is None else y + shift.view(1, -1, 1, 1), act)
        used['conv2x2'] = used.get('conv2x2', 0) + 1
        if pool_partial is not None:                                                         # sums over blocks of 16 pixels
            flat = y.flatten(2)
            flat = F.pad(flat, (0, pool_partial.shape[2] * 16 - flat.shape[2]))
            pool_partial.copy_(flat.view(*pool_partial.shape, 16).sum(-1))
        return y

    def gemm_split_up2(sw, x, shift=None, act=0, out=None):
        y = F.conv2d(x, weight_of(sw)[:, :, None, None])
        y = act_of(y if shift is None else y + shift.view(1, -1, 1, 1), act)
        used['up2'] = used.get('up2', 0) + 1
        return out.copy_(F.interpolate(y, scale_factor=2, mode='nearest'))

    def pooled_shift(pool_partial, pixels, wb, shift):
        used['pooled_shift'] = used.get('pooled_shift', 0) + 1
        return shift + wb @ (pool_partial[0].sum(1) / pixels)

    monkeypatch.setattr(HF, 'gemm_split', gemm_split)
    monkeypatch.setattr(HF, 'gemm_split_conv2x2', gemm_split_conv2x2)
    monkeypatch.setattr(HF, 'gemm_split_up2', gemm_split_up2)
    monkeypatch.setattr(HF, 'pooled_shift', pooled_shift)
    monkeypatch.setattr(HF, 'mbconv_expand_dw', expand_dw)
    monkeypatch.setattr(HF, 'depthwise_conv_bn_act', dw)
    monkeypatch.setattr(HF, 'se_gate', se_gate)
    monkeypatch.setattr(HF, 'pointwise_conv', pointwise)
    monkeypatch.setattr(HF, 'affine_act_', affine)

    stock = fill_by_name(configs.build(config).eval(), seed=3)
    fused = copy.deepcopy(stock)
    prepare_for_inference(fused, fold_bn=False, fused_depthwise=True, split_gemm=split)
    with pytest.raises(RuntimeError):                     # not idempotent by design: refuses a second application
        prepare_for_inference(fused, fold_bn=False, fused_depthwise=True)
    bb = fused.backbone
    deferred = [b._fused_dw.defer_shift for b in bb._blocks]
    assert sum(deferred) >= len(deferred) - 3 and not deferred[0]     # block 0 feeds a depthwise conv directly
    for size, batch in (((128, 256), 1), ((64, 128), 2)):             # 128x256: first blocks take the "MFMA" route
        x = torch.rand(batch, 3, *size, generator=G(1031))
        with torch.no_grad():
            ref = stock.backbone(x)
            t = F.silu(bb._bn0(bb._conv_stem(x)))
            feats = []
            for idx, blk in enumerate(bb._blocks):
                t = blk._fused_dw(t, blk)
                if bb._res_feat_mask[idx]:
                    key = str(len(feats))
                    feats.append(bb._fused_fc[key](t) if key in bb._fused_fc else t.clone())
            feats.append(bb._fused_head(t))
        assert len(feats) == len(ref)
        for a, b in zip(ref, feats):
            assert rel_err(b, a) < 2e-5
        # the opt-in route is taken (or not at all): expand, gated project and in-place skip accumulation all go through it
        assert (used['n'] > 20 and used['gated'] > 10 and used['accumulated'] > 5) if split else used['n'] == 0
        if batch == 1 and getattr(fused.weight_mapper, '_fused', None) is not None:
            # the context head without its concatenations (FusedContextHead) == the stock head
            with torch.no_grad():
                assert rel_err(fused.weight_mapper._fused(feats[-1].contiguous()), stock.weight_mapper(ref[-1])) < 2e-5
            if split and feats[-1].shape[3] % 4 == 0:
                assert used.get('conv2x2', 0) >= 1              # the head's 2x2 / stride-2 down blocks take the windowed GEMM,
                assert used.get('pooled_shift', 0) >= 1 and used.get('up2', 0) >= 1     # its pool and its upsample ride along


def test_bank_slices_autograd_cpu():
    """autograd.BankSlices (host logic, no GPU): the three column ranges of an inverted residual's bank as views whose gradients come
    back as ONE concatenation -- equal to plain slicing, unused ranges and the unused tail included."""
    from hyperseg_amd.autograd import BankSlices
    g = torch.Generator().manual_seed(1)
    bank = torch.randn(6, 23, generator=g)
    a, b = bank.clone().requires_grad_(True), bank.clone().requires_grad_(True)
    x1, x2, x3 = BankSlices.apply(a, 5, 11, 20)
    y1, y2, y3 = b[:, :5], b[:, 5:11], b[:, 11:20]
    assert torch.equal(x1, y1) and torch.equal(x2, y2) and torch.equal(x3, y3) and x2.data_ptr() == a.data_ptr() + 5 * 4
    r = torch.randn(6, 9, generator=g)
    ((x1 ** 2).sum() + (x3 * r).sum()).backward()              # the middle range takes no part
    ((y1 ** 2).sum() + (y3 * r).sum()).backward()
    assert torch.equal(a.grad, b.grad) and float(a.grad[:, 5:11].abs().max()) == 0.0 and float(a.grad[:, 20:].abs().max()) == 0.0


def test_patch_ir_routes():
    """hs_patch_ir_route (host only): which kernel a fused inverted-residual level gets.  The f16-split matrix-core kernel is
    chosen by RANGES of channel counts, so every BASELINE level 4, CamVid-L's 6-level model (20-class variant included) and odd
    shapes all get it; level-3 shapes (8 x 8 patches) keep the exact-f32 matrix-core kernel; 'f32' math never takes the split
    form; beyond 16 + 16 channels the generic kernel is what is left -- and the query says so instead of hiding it."""
    from hyperseg_amd import functional as HF
    cases = {'M level 4': (((1, 256, 512), 16, 16, (16, 32), 68, 19), 'split_mfma', 'split_mfma', 'f32_mfma'),
             'S level 4': (((1, 384, 768), 16, 8, (24, 48), 52, 19), 'split_mfma', 'split_mfma', 'f32_mfma'),
             'CamVid-S level 4': (((1, 288, 384), 4, 16, (18, 24), 44, 12), 'split_mfma', 'split_mfma', 'f32_mfma'),
             'M level 3': (((1, 128, 256), 6, 16, (16, 32), 48, 16), 'f32_mfma', 'f32_mfma', 'f32_mfma'),
             'CamVid-L level 5, 20 classes': (((1, 768, 1024), 3, 16, (24, 32), 42, 20), 'split_mfma', 'split_mfma', 'generic'),
             # round 6: CamVid-L's own two blocks are in the exact-f32 kernel's table, and AUTO -- "the faster form" -- takes it for
             # narrow blocks (<= 4 skip channels) in launches of more than 512 regions (measured: include/hyperseg_hip.h hs_ir_math)
             'CamVid-L level 5': (((1, 768, 1024), 3, 16, (24, 32), 42, 12), 'f32_mfma', 'split_mfma', 'f32_mfma'),
             'CamVid-L level 4': (((1, 384, 512), 4, 16, (24, 32), 44, 16), 'f32_mfma', 'split_mfma', 'f32_mfma'),
             'odd channel counts': (((1, 64, 96), 5, 7, (4, 6), 30, 9), 'split_mfma', 'split_mfma', 'generic'),
             '20 skip channels': (((1, 64, 64), 20, 16, (4, 4), 76, 19), 'generic', 'generic', 'generic')}
    for name, (args, auto, split, f32) in cases.items():
        assert HF.patch_ir_route(*args, math='auto') == auto, name
        assert HF.patch_ir_route(*args, math='split') == split, name
        assert HF.patch_ir_route(*args, math='f32') == f32, name
    with pytest.raises(Exception):
        HF.patch_ir_route((1, 100, 100), 4, 4, (3, 3), 16, 8)              # 100 % 3 != 0


def test_c_abi_exports_match_header():
    """The in-tree shared library loads without a GPU and exports every entry point include/hyperseg_hip.h declares (and the
    ctypes binding knows exactly that set); no compute call is made."""
    import ctypes
    import os
    import re
    from conftest import REPO
    from hyperseg_amd import _hip
    header = open(os.path.join(REPO, 'include', 'hyperseg_hip.h')).read()
    header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)
    declared = set(re.findall(r'\b(hs_[a-z0-9_]+)\s*\(', header))
    assert len(declared) >= 20
    lib = ctypes.CDLL(_hip._LIB_PATH)
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, f'declared in the header but not exported by the library: {missing}'
    assert declared == set(_hip.EXPORTS), sorted(declared ^ set(_hip.EXPORTS))
    assert lib.hs_version() == 1
    lib.hs_build_info.restype = ctypes.c_char_p
    assert b'gfx950' in lib.hs_build_info()


def test_k1_chain_plan_on_the_host():
    """hs_k1_chain_workspace is host logic only (no device call): the shapes of the v1_0 configurations are accepted, the workspace is the
    header word block + one 8-byte granule per level-0 / level-1 output, and shapes past the kernel's LDS-DMA / thread budgets are refused
    with HS_ERR_UNSUPPORTED (the caller then issues three hs_patch_conv_fwd launches)."""
    import ctypes as C
    from hyperseg_amd import _hip

    def plan(batch, fh, fw, chans, n=3):
        arr = (_hip.K1LevelC * 3)()
        dummy = C.c_void_p(4096)                           # never dereferenced by the planner: alignment is all it looks at
        prev = 0
        for l, (cs, co) in enumerate(chans):
            arr[l].skip, arr[l].c_skip, arr[l].bank, arr[l].c_out = dummy, cs, dummy, co
            arr[l].ld = (co * (2 + cs + prev) + 3) & ~3
            arr[l].scale = arr[l].shift = None
            prev = co
        return int(_hip.lib.hs_k1_chain_workspace(batch, fh, fw, arr, n))
    m = [(80, 64), (28, 32), (10, 16)]                      # HyperSeg-M / CamVid-S levels 0-2: 82 -> 64, 94 -> 32, 44 -> 16
    assert plan(1, 16, 32, m) == 256 + 8 * 512 * (64 + 4 * 32)
    assert plan(1, 24, 48, [(128, 32), (28, 16), (8, 8)]) == 256 + 8 * 1152 * (32 + 4 * 16)        # HyperSeg-S
    assert plan(1, 16, 32, m, n=2) == -3                     # three levels or nothing
    assert plan(1, 16, 32, [(80, 96), (28, 32), (10, 16)]) == -3          # c_out > 64
    assert plan(1, 16, 32, [(120, 64), (28, 32), (10, 16)]) == -3         # level-0 bank past 24 KB
    assert plan(1, 16, 32, [(80, 64), (70, 32), (10, 16)]) == -3          # 70 skip channels x 4 pixels > 256 gather lanes
    assert plan(0, 16, 32, m) == -1

    def plan_ir(batch, fh, fw, chans, cs3, hid, co3):
        arr = (_hip.K1LevelC * 3)()
        dummy = C.c_void_p(4096)
        prev = 0
        for l, (cs, co) in enumerate(chans):
            arr[l].skip, arr[l].c_skip, arr[l].bank, arr[l].c_out = dummy, cs, dummy, co
            arr[l].ld = (co * (2 + cs + prev) + 3) & ~3
            prev = co
        ir = _hip.ChainIrLevelC()
        ir.skip, ir.c_skip, ir.bank, ir.hidden, ir.c_out = dummy, cs3, dummy, hid, co3
        cin = 2 + cs3 + prev
        ir.ld = (cin * hid + 9 * hid + hid * co3 + 3) & ~3
        return int(_hip.lib.hs_decoder_chain_workspace(batch, fh, fw, arr, 3, C.byref(ir)))
    # with the first inverted-residual level (HyperSeg-M / CamVid-S level 3: 24 -> 48 -> 16 on 8 x 8 patches): + 16 granules x 16 channels per cell
    assert plan_ir(1, 16, 32, m, 6, 48, 16) == 256 + 8 * 512 * (64 + 4 * 32 + 16 * 16)
    assert plan_ir(1, 16, 32, m, 6, 46, 16) == -3            # hidden not a multiple of 4
    assert plan_ir(1, 16, 32, m, 6, 96, 16) == -3            # hidden > 64
    assert plan_ir(1, 16, 32, m, 8, 48, 16) == -3            # 800 skip-halo gathers > 768


def test_se_tail_plan_on_the_host():
    """csrc/hs_se_tail.h's split of a block's squeeze-excite gate over the last workgroups of its pooling launch (host-side plan,
    no launch): every MBConv block of EfficientNet-B1 at HyperSeg-M's 1024x512 is covered, <= 32 tails, <= 2048 partial granules
    and <= 64 channels each; shapes outside the plan return 0 (the caller keeps hs_se_gate_fwd)."""
    from hyperseg_amd import _hip
    lib = _hip.lib
    # (channels, squeezed, partials per channel, workgroups per batch element)
    blocks = [(32, 8, 128, 32 * 128), (16, 4, 128, 16 * 128), (96, 4, 256, 256), (144, 6, 128, 256), (144, 6, 64, 192), (240, 10, 64, 256),
              (240, 10, 16, 240), (480, 20, 2, 960), (672, 28, 2, 1344), (672, 28, 1, 672), (1152, 48, 1, 1152), (1920, 80, 1, 1920)]
    for c, csq, nblk, wgs in blocks:
        t = lib.hs_se_tail_tails(c, csq, nblk, wgs)
        assert 1 <= t <= 32, (c, csq, nblk, t)
        per = -(-c // t)
        assert per <= 64 + 3 and per * nblk <= 2048 + 3 * nblk, (c, nblk, t)
        words = lib.hs_se_tail_workspace(2, c, csq, nblk, wgs) // 8
        assert words == 2 + 1 + 2 * t * csq + 2 * c * nblk
    assert lib.hs_se_tail_tails(64, 200, 1, 64) == 0            # > 96 squeezed channels
    assert lib.hs_se_tail_tails(64, 8, 1024, 64 * 1024) == 0    # > 512 partials per channel
    assert lib.hs_se_tail_tails(640, 8, 1, 4) == 0              # fewer workgroups than tails
    assert lib.hs_se_tail_workspace(1, 64, 200, 1, 64) == 0


def test_training_host_logic_without_a_gpu():
    """Round-5 training helpers, the parts that run without a device: the shared bank-gradient buffer's bookkeeping (autograd._BankGradBuffer:
    its views tile one tensor, it recognises its own views and nothing else), the one-launch Adam's refusals (CPU parameters are NOT
    stepped by some fallback: NotImplementedError) and hs_adam_blocks' workgroup count."""
    import ctypes as C
    from hyperseg_amd import _hip, autograd as HA
    from hyperseg_amd.training import Adam
    buf = HA._BankGradBuffer((5, 12), ((0, 3), (3, 7), (7, 10)))
    v = [buf.view(i, torch.device('cpu')) for i in range(3)]
    assert [tuple(t.shape) for t in v] == [(5, 3), (5, 4), (5, 3)] and all(t.stride() == (12, 1) for t in v)
    assert float(buf.buf[:, 10:].abs().max()) == 0.0                    # the pad columns are zeroed once
    for i, t in enumerate(v):
        t.fill_(i + 1.0)
    assert buf.owns(v) and torch.equal(buf.buf[:, :10], torch.tensor([1.0] * 3 + [2.0] * 4 + [3.0] * 3).expand(5, 10))
    assert not buf.owns([v[0], v[1], None]) and not buf.owns([v[0], v[1], torch.zeros(5, 3)]) and not buf.owns([v[1], v[0], v[2]])
    assert not HA._BankGradBuffer((5, 12), ((0, 3), (3, 7), (7, 10))).owns(v)          # another buffer's views
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    with pytest.raises(NotImplementedError):
        Adam([p], lr=1e-3).step()
    with pytest.raises(ValueError):
        Adam([p], lr=1e-3, betas=(1.0, 0.999))
    numel = (C.c_int64 * 4)(1, 1024, 1025, 4216 * 80)
    assert _hip.lib.hs_adam_blocks(numel, 4) == 1 + 1 + 2 + 330
    assert _hip.lib.hs_adam_blocks(numel, 0) == 0 and _hip.lib.hs_adam_blocks((C.c_int64 * 1)(0), 1) == 0


def test_the_product_never_imports_the_oracle():
    """The oracle is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.  Nothing under
    hyperseg_amd/ (the package, incl. hyperseg_amd/benchlib: bench.py's other legs) may -- a product path that routes through the CPU oracle
    would void every parity claim -- and in bench.py the import sits inside cpu_baseline() only."""
    import ast
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def oracle_imports(path):
        tree = ast.parse(open(path).read())
        hits = []
        for node in ast.walk(tree):
            if isinstance(node, ast.Import) and any(a.name.split('.')[0] == 'oracle' for a in node.names):
                hits.append(node.lineno)
            if isinstance(node, ast.ImportFrom) and node.level == 0 and (node.module or '').split('.')[0] == 'oracle':
                hits.append(node.lineno)
        return tree, hits
    for d, _, files in os.walk(os.path.join(root, 'hyperseg_amd')):
        for f in files:
            if f.endswith('.py'):
                assert not oracle_imports(os.path.join(d, f))[1], f'{os.path.join(d, f)} imports oracle/'
    tree, hits = oracle_imports(os.path.join(root, 'bench.py'))
    cb = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == 'cpu_baseline')
    assert hits and all(cb.lineno <= h <= cb.end_lineno for h in hits), 'bench.py may import oracle/ inside cpu_baseline() only'
