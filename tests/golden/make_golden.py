"""Generate the golden fixtures in tests/golden/*.npz by IMPORTING THE REFERENCE.

Run in the build container only (needs /root/reference; the reference never travels):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Each fixture holds seeded inputs, the reference module's parameters (state-dict key names) and
the reference's outputs.  Fixtures are data: no reference source is stored.  The reference
defines no tests of its own for this path; its only known-answer
(meta_conv.py:233-242 -> 9.0) is captured in ``meta_conv_known_answer``.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, '/root/reference')
sys.path.insert(0, REPO)
sys.modules.setdefault('ffmpeg', types.ModuleType('ffmpeg'))   # utils.py:9 imports it, never used here

from hyperseg.models.layers.meta_conv import MetaConv2d                                     # noqa: E402
from hyperseg.models.layers.meta_patch import MetaPatchConv2d, make_meta_patch_conv2d_block  # noqa: E402
from hyperseg.models.layers.meta_sequential import MetaSequential                           # noqa: E402
import hyperseg.models.hyperseg_v1_0 as v1                                                   # noqa: E402
import hyperseg.models.hyperseg_v1_0_unify as vu                                             # noqa: E402
import hyperseg.models.hyperseg_v0_1 as v0                                                   # noqa: E402
from oracle import hyperseg_oracle as O                                                      # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(8)


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB')


def randomize_bn(module, gen):
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            for k, v in O.synth_bn(gen, m.num_features).items():
                getattr(m, k).copy_(v)


def sd(module, prefix='p.'):
    return {prefix + k: v for k, v in module.state_dict().items() if 'num_batches_tracked' not in k}


# ------------------------------------------------------------------------------ a2 MetaConv2d
def gen_meta_conv():
    m = MetaConv2d(3, 3, 3, padding=1, groups=3)
    x = torch.ones(4, 3, 64, 64)
    x[0::2] = 0.
    w = torch.ones(4, m.hyper_params)
    w[0::2] = 0.
    save('meta_conv_known_answer', out_max=m(x, w).max())

    g = torch.Generator().manual_seed(1)
    cases = [
        dict(cin=5, cout=7, k=1, padding=0, groups=1, mode='zeros'),
        dict(cin=6, cout=6, k=3, padding=1, groups=6, mode='zeros'),
        dict(cin=6, cout=4, k=3, padding=1, groups=2, mode='reflect'),
        dict(cin=4, cout=8, k=3, padding=0, groups=1, mode='zeros'),
        dict(cin=3, cout=3, k=3, padding=1, groups=1, mode='replicate'),
    ]
    arrs = {'n': len(cases)}
    for i, c in enumerate(cases):
        m = MetaConv2d(c['cin'], c['cout'], c['k'], padding=c['padding'], groups=c['groups'], padding_mode=c['mode'])
        x = torch.randn(3, c['cin'], 9, 11, generator=g)
        w = torch.randn(3, int(m.hyper_params), generator=g)
        arrs.update({f'{i}.x': x, f'{i}.w': w, f'{i}.y': m(x, w),
                     f'{i}.cfg': np.array([c['cin'], c['cout'], c['k'], c['padding'], c['groups']]),
                     f'{i}.mode': c['mode']})
    save('meta_conv2d', **arrs)


def gen_meta_conv_general():
    """MetaConv2d outside "same" padding / stride 1 / dilation 1 (meta_conv.py:141-186), incl. the three examples of the
    reference's own docstring (meta_conv.py:127-131, at a smaller size)."""
    g = torch.Generator().manual_seed(21)
    cases = [
        dict(cin=16, cout=33, k=3, stride=2, padding=0, dilation=1, groups=1, mode='zeros', size=(13, 17)),
        dict(cin=16, cout=33, k=(3, 5), stride=(2, 1), padding=(4, 2), dilation=1, groups=1, mode='zeros', size=(14, 19)),
        dict(cin=16, cout=33, k=(3, 5), stride=(2, 1), padding=(4, 2), dilation=(3, 1), groups=1, mode='zeros', size=(15, 21)),
        dict(cin=4, cout=8, k=3, stride=1, padding=0, dilation=1, groups=1, mode='zeros', size=(9, 11)),        # "valid"
        dict(cin=6, cout=4, k=3, stride=2, padding=2, dilation=2, groups=2, mode='reflect', size=(10, 12)),
        dict(cin=5, cout=5, k=(1, 3), stride=(1, 2), padding=(0, 3), dilation=(1, 2), groups=5, mode='replicate', size=(7, 9)),
        dict(cin=3, cout=6, k=(5, 2), stride=3, padding=(3, 1), dilation=1, groups=3, mode='circular', size=(11, 8)),
        dict(cin=8, cout=2, k=2, stride=2, padding=0, dilation=1, groups=1, mode='zeros', size=(8, 8)),
        dict(cin=2, cout=3, k=3, stride=1, padding=2, dilation=1, groups=1, mode='reflect', size=(6, 7)),       # "full"-ish
    ]
    arrs = {'n': len(cases)}
    for i, c in enumerate(cases):
        m = MetaConv2d(c['cin'], c['cout'], c['k'], stride=c['stride'], padding=c['padding'], dilation=c['dilation'],
                       groups=c['groups'], padding_mode=c['mode'])
        x = torch.randn(2, c['cin'], *c['size'], generator=g)
        w = torch.randn(2, int(m.hyper_params), generator=g)
        arrs.update({f'{i}.x': x, f'{i}.w': w, f'{i}.y': m(x, w),
                     f'{i}.cfg': np.array([c['cin'], c['cout'], *m.kernel_size, *m.stride, *m.padding, *m.dilation, c['groups']]),
                     f'{i}.mode': c['mode']})
    save('meta_conv2d_general', **arrs)


# ------------------------------------------------------------------------------ a3 MetaPatchConv2d
def gen_meta_patch():
    g = torch.Generator().manual_seed(2)
    cases = [
        dict(cin=5, cout=7, k=1, groups=1, b=2, grid=(3, 4), patch=(2, 4)),
        dict(cin=11, cout=3, k=1, groups=1, b=1, grid=(2, 2), patch=(1, 1)),
        dict(cin=6, cout=4, k=3, groups=1, b=2, grid=(3, 4), patch=(4, 2)),
        dict(cin=6, cout=6, k=3, groups=6, b=2, grid=(2, 3), patch=(8, 8)),
        dict(cin=4, cout=6, k=3, groups=2, b=1, grid=(4, 2), patch=(2, 2)),
        dict(cin=8, cout=8, k=1, groups=4, b=2, grid=(2, 2), patch=(4, 4)),
    ]
    arrs = {'n': len(cases)}
    for i, c in enumerate(cases):
        m = MetaPatchConv2d(c['cin'], c['cout'], c['k'], padding=c['k'] // 2, groups=c['groups'])
        h, w = c['grid'][0] * c['patch'][0], c['grid'][1] * c['patch'][1]
        x = torch.randn(c['b'], c['cin'], h, w, generator=g)
        wt = torch.randn(c['b'], int(m.hyper_params), *c['grid'], generator=g)
        arrs.update({f'{i}.x': x, f'{i}.w': wt, f'{i}.y': m(x, wt),
                     f'{i}.cfg': np.array([c['cin'], c['cout'], c['k'], c['groups']])})
    # block with BN + ReLU (make_meta_patch_conv2d_block, meta_patch.py:228-257), eval mode
    blk = make_meta_patch_conv2d_block(6, 5, 1).eval()
    randomize_bn(blk, g)
    x = torch.randn(2, 6, 8, 12, generator=g)
    wt = torch.randn(2, blk.hyper_params, 2, 3, generator=g)
    arrs.update({'blk.x': x, 'blk.w': wt, 'blk.y': blk(x, wt), **sd(blk, 'blk.p.')})
    # shape smoke of meta_patch.py main(): x 2x10x256x256, w ones 2xhpx8x8 -> [2,20,256,256]
    m = MetaPatchConv2d(10, 20, 3, padding=0)   # main() wraps MetaConv2d(kernel_size=3) with padding=0
    save('meta_patch_conv2d', **arrs)


# ------------------------------------------------------------------------------ a1 MetaSequential
def gen_meta_sequential():
    g = torch.Generator().manual_seed(3)
    seq = MetaSequential(MetaPatchConv2d(4, 6, 1), torch.nn.ReLU(), MetaPatchConv2d(6, 3, 3, padding=1)).eval()
    x = torch.randn(2, 4, 6, 8, generator=g)
    hp0, hp1 = int(seq[0].hyper_params), int(seq[2].hyper_params)
    w_cat = torch.randn(2, hp0 + hp1, 3, 2, generator=g)
    y_tensor = seq(x, w_cat)
    y_list = seq(x, [w_cat[:, :hp0].contiguous(), w_cat[:, hp0:].contiguous()])
    # clamped slice: tensor with MORE channels than hyper_params -> the tail is ignored
    w_long = torch.cat([w_cat, torch.randn(2, 5, 3, 2, generator=g)], dim=1)
    y_long = seq(x, w_long)
    save('meta_sequential', x=x, w=w_cat, y_tensor=y_tensor, y_list=y_list, y_long=y_long,
         ranges=np.array(seq._ranges), hp=np.array([hp0, hp1]))


# ------------------------------------------------------------------------------ a4/a5/a9 v1_0 hyper patch modules
def gen_hyper_patch():
    g = torch.Generator().manual_seed(4)
    arrs = {}
    # a4: HyperPatchNoPadding with next_multiply padding (hp=35 not divisible by G=4 -> 36 rows)
    m = v1.HyperPatchNoPadding(7, 5, 1)
    m.init_signal2weights(16, 3, 4)
    m.signal2weights.weight.copy_(torch.randn(m.signal2weights.weight.shape, generator=g))
    x = torch.randn(2, 7, 6, 8, generator=g)
    s = torch.randn(2, 24, 3, 4, generator=g).clamp(min=0)
    arrs.update({'np.x': x, 'np.s': s, 'np.y': m(x, s), 'np.wt': m.apply_signal2weights(s),
                 'np.w_s2w': m.signal2weights.weight, 'np.cfg': np.array([7, 5, 16, 3, 4, int(m.hyper_params)])})
    # a5: HyperPatchConv2d k=3 reflect (not instantiated by BASELINE configs, kept for API parity)
    m = v1.HyperPatchConv2d(4, 6, 3, padding=1)
    m.init_signal2weights(8, 0, 2)
    m.signal2weights.weight.copy_(torch.randn(m.signal2weights.weight.shape, generator=g))
    x = torch.randn(2, 4, 6, 8, generator=g)
    s = torch.randn(2, 8, 3, 4, generator=g).clamp(min=0)
    arrs.update({'pc.x': x, 'pc.s': s, 'pc.y': m(x, s), 'pc.w_s2w': m.signal2weights.weight,
                 'pc.cfg': np.array([4, 6, 8, 0, 2, int(m.hyper_params)])})
    # make_hyper_patch_conv2d_block (BN + ReLU) with the signal arriving through MetaSequential's
    # clamped slice (Appendix D-2): s has fewer channels than hp
    blk = v1.make_hyper_patch_conv2d_block(7, 5, 1).eval()
    blk[0].init_signal2weights(16, 0, 4)
    blk[0].signal2weights.weight.copy_(torch.randn(blk[0].signal2weights.weight.shape, generator=g))
    randomize_bn(blk, g)
    x = torch.randn(2, 7, 6, 8, generator=g)
    s = torch.randn(2, 20, 3, 4, generator=g).clamp(min=0)
    arrs.update({'blk.x': x, 'blk.s': s, 'blk.y': blk(x, s), **sd(blk, 'blk.p.'),
                 'blk.cfg': np.array([7, 5, 16, 0, 4, int(blk[0].hyper_params)])})
    save('hyper_patch_v1', **arrs)


# ------------------------------------------------------------------------------ a6 Op C
def gen_ir_v1():
    g = torch.Generator().manual_seed(5)
    cases = [
        dict(cin=5, cout=3, er=2, b=2, grid=(3, 4), patch=(4, 2), cs=8, idx=2, grp=4),
        dict(cin=6, cout=6, er=2, b=1, grid=(2, 2), patch=(8, 8), cs=12, idx=0, grp=2),    # residual connect
        dict(cin=7, cout=4, er=1.5, b=2, grid=(2, 3), patch=(2, 2), cs=6, idx=0, grp=3),
        dict(cin=4, cout=5, er=2, b=1, grid=(1, 1), patch=(16, 16), cs=4, idx=0, grp=1),
    ]
    arrs = {'n': len(cases)}
    for i, c in enumerate(cases):
        m = v1.HyperPatchInvertedResidual(c['cin'], c['cout'], 3, expand_ratio=c['er']).eval()
        m.init_signal2weights(c['cs'], c['idx'], c['grp'])
        m.signal2weights.weight.copy_(torch.randn(m.signal2weights.weight.shape, generator=g) * 0.5)
        randomize_bn(m, g)
        h, w = c['grid'][0] * c['patch'][0], c['grid'][1] * c['patch'][1]
        x = torch.randn(c['b'], c['cin'], h, w, generator=g)
        s = torch.randn(c['b'], c['idx'] + c['cs'] + 1, *c['grid'], generator=g).clamp(min=0)
        arrs.update({f'{i}.x': x, f'{i}.s': s, f'{i}.y': m(x, s), f'{i}.wt': m.apply_signal2weights(s),
                     f'{i}.cfg': np.array([c['cin'], c['cout'], m.hidden_dim, c['cs'], c['idx'], c['grp'],
                                           int(m.hyper_params)]), **sd(m, f'{i}.p.')})
    # unify flavour: weights arrive directly (hyperseg_v1_0_unify.py:312-389)
    m = vu.HyperPatchInvertedResidual(5, 3, 3, expand_ratio=2).eval()
    randomize_bn(m, g)
    x = torch.randn(2, 5, 8, 12, generator=g)
    wt = torch.randn(2, int(m.hyper_params), 2, 3, generator=g) * 0.5
    arrs.update({'u.x': x, 'u.wt': wt, 'u.y': m(x, wt), 'u.cfg': np.array([5, 3, m.hidden_dim]), **sd(m, 'u.p.')})
    save('inverted_residual_v1', **arrs)


# ------------------------------------------------------------------------------ a7 Op D
def gen_ir_v0():
    g = torch.Generator().manual_seed(6)
    cases = [dict(cin=5, cout=3, b=2, grid=(3, 4), patch=(4, 2)),
             dict(cin=6, cout=6, b=1, grid=(2, 2), patch=(8, 8)),
             dict(cin=4, cout=7, b=2, grid=(2, 2), patch=(1, 1))]
    arrs = {'n': len(cases)}
    for i, c in enumerate(cases):
        m = v0.HyperPatchInvertedResidual(c['cin'], c['cout'], 3, expand_ratio=2).eval()
        randomize_bn(m, g)
        h, w = c['grid'][0] * c['patch'][0], c['grid'][1] * c['patch'][1]
        x = torch.randn(c['b'], c['cin'], h, w, generator=g)
        wt = torch.randn(c['b'], int(m.hyper_params), *c['grid'], generator=g) * 0.5
        arrs.update({f'{i}.x': x, f'{i}.wt': wt, f'{i}.y': m(x, wt),
                     f'{i}.cfg': np.array([c['cin'], c['cout'], c['cin'] * 2]), **sd(m, f'{i}.p.')})
    save('inverted_residual_v0', **arrs)


# ------------------------------------------------------------------------------ divide_feature
def gen_divide_feature():
    arrs = {}
    for name in ('M', 'S', 'Sc'):
        plan = O.config_plan(name)
        targets = [sw['hp'] for sw in plan['s2w']]
        wg = O.CONFIGS[name]['weight_groups']
        arrs[f'{name}.targets'] = np.array(targets)
        arrs[f'{name}.min_unit'] = max(wg)
        arrs[f'{name}.split'] = v1.divide_feature(1280, targets, min_unit=max(wg))
    rng = np.random.RandomState(0)
    for i in range(12):
        n = rng.randint(2, 7)
        targets = [int(t) for t in rng.randint(50, 9000, size=n)]
        if i % 3 == 0:
            targets[1] = targets[0]
        mu = int(rng.choice([4, 8, 16, 32]))
        total = mu * int(rng.randint(n + 2, 80))
        arrs[f'r{i}.targets'] = np.array(targets)
        arrs[f'r{i}.min_unit'] = mu
        arrs[f'r{i}.total'] = total
        arrs[f'r{i}.split'] = v1.divide_feature(total, targets, min_unit=mu)
        arrs[f'r{i}.legacy'] = v0.divide_feature_legacy(total, targets, min_unit=mu)
    save('divide_feature', **arrs)


# ------------------------------------------------------------------------------ a8 decoders (tiny + full)
def ref_decoder(name_or_cfg):
    c = O.CONFIGS[name_or_cfg] if isinstance(name_or_cfg, str) else name_or_cfg
    if c['variant'] == 'v1_0':
        d = v1.MultiScaleDecoder(c['feat'], c['signal'], c['num_classes'], c['kernel_sizes'], 1,
                                 c['level_channels'], expand_ratio=c['expand_ratio'],
                                 weight_groups=list(c['weight_groups']))
    elif c['variant'] == 'unify':
        d = vu.MultiScaleDecoder(c['feat'], c['signal'], c['num_classes'], c['kernel_sizes'], 1,
                                 c['level_channels'], expand_ratio=c['expand_ratio'],
                                 weight_groups=list(c['weight_groups']), unify_level=c['unify_level'])
    else:
        d = v0.MultiScaleDecoder(c['feat'], 3, c['num_classes'], c['kernel_sizes'], 1,
                                 expand_ratio=c['expand_ratio'])
    return d.eval()


def load_params(dec, params):
    missing, unexpected = dec.load_state_dict(params, strict=False)
    missing = [k for k in missing if 'num_batches_tracked' not in k and not k.startswith('coord')]
    assert not missing and not unexpected, (missing, unexpected)


TINY = {
    't_v1_0': dict(variant='v1_0', size=(64, 96), feat=[3, 4, 3, 5, 6, 8], signal=48, num_classes=5,
                   kernel_sizes=[1, 1, 1, 3, 3], level_channels=[8, 6, 4, 4, 4], expand_ratio=2,
                   weight_groups=[4, 2, 2, 4, 2]),
    't_unify': dict(variant='unify', size=(64, 96), feat=[3, 4, 3, 5, 6, 8], signal=64, num_classes=5,
                    kernel_sizes=[1, 1, 1, 3, 3], level_channels=[8, 6, 4, 4, 4], expand_ratio=2,
                    weight_groups=[4, 2, 2, 4, 2], unify_level=4),
    't_v0_1': dict(variant='v0_1', size=(64, 64), feat=[3, 2, 3, 4, 5, 6], signal=16, num_classes=4,
                   kernel_sizes=[1, 1, 3, 3, 3, 3], expand_ratio=2),
}


def gen_decoders():
    for name, cfg in TINY.items():
        O.CONFIGS[name] = cfg
        plan = O.config_plan(name)
        params = O.synth_decoder_params(plan, seed=7)
        dec = ref_decoder(cfg)
        load_params(dec, params)
        x, sw = O.synth_decoder_inputs(name, batch=2, seed=7)
        y = dec(x, sw)
        arrs = {f'x{i}': t for i, t in enumerate(x)}
        if isinstance(sw, list):
            arrs.update({f'w{i}': t for i, t in enumerate(sw)})
        else:
            arrs['s'] = sw
        arrs.update({'p.' + k: v for k, v in params.items()})
        arrs['y'] = y
        # reference plan facts
        if cfg['variant'] != 'v0_1':
            arrs['hyper_params'] = np.array(dec.param_groups)
        save('decoder_' + name, **arrs)

    # full BASELINE shapes: seeds + SHA-256 of the argmax mask + strided logits sample
    arrs = {}
    for name in ('M', 'S', 'Sc', 'L', 'Lc'):
        plan = O.config_plan(name)
        params = O.synth_decoder_params(plan, seed=0)
        dec = ref_decoder(name)
        load_params(dec, params)
        batch = 4 if name == 'L' else 1      # L: the per-GPU shard of BASELINE config 4 (bs 32 over 8 GPUs)
        x, sw = O.synth_decoder_inputs(name, batch=batch, seed=0)
        y = dec(x, sw)
        # NB: the minimum top-2 margin over ~0.5 M pixels is ~1 ulp, so a bit-identical mask cannot
        # be demanded of ANY re-ordered fp32 summation; masks must agree wherever the reference's
        # own margin exceeds the fp32 error bound (tests use 1e-4, >100x the observed error).
        top2 = y.topk(2, dim=1).values
        margin = top2[:, 0] - top2[:, 1]
        mask = y.argmax(1)
        arrs[f'{name}.logits_sample'] = y[:, :, 3::37, 5::41].contiguous()
        arrs[f'{name}.mask_sample'] = mask[:, 3::37, 5::41].to(torch.uint8)
        arrs[f'{name}.margin_sample'] = margin[:, 3::37, 5::41].contiguous()
        arrs[f'{name}.logits_absmax'] = y.abs().max()
        arrs[f'{name}.n_margin_below_1e-4'] = int((margin < 1e-4).sum())
        arrs[f'{name}.batch'] = batch
        # provenance: how close the oracle is to the reference on the full tensor, at generation time
        fn = {'v1_0': O.decoder_v1_0, 'unify': O.decoder_unify, 'v0_1': O.decoder_v0_1}[plan['variant']]
        yo = fn(plan, params, x, sw)
        flips = (yo.argmax(1) != mask)
        arrs[f'{name}.oracle_maxabs_err'] = (yo - y).abs().max()
        arrs[f'{name}.oracle_flips'] = int(flips.sum())
        arrs[f'{name}.oracle_flips_margin_gt_1e-4'] = int((flips & (margin > 1e-4)).sum())
        print(name, tuple(y.shape), 'absmax', float(y.abs().max()), 'oracle max err', float((yo - y).abs().max()),
              'flips', int(flips.sum()), 'flips@margin>1e-4', int((flips & (margin > 1e-4)).sum()),
              'n(margin<1e-4)', int((margin < 1e-4).sum()))
    save('decoder_full_configs', **arrs)


# ------------------------------------------------------------------------------ training step (config 5 contract)
def gen_train():
    """Train-mode forward + backward of the tiny decoders in the REFERENCE: loss = sum(y * R); gradients w.r.t. every
    pyramid input, the signal / weights, signal2weights.weight, BN gamma/beta; BN running stats after the step."""
    for name, cfg in TINY.items():
        O.CONFIGS[name] = cfg
        plan = O.config_plan(name)
        params = O.synth_decoder_params(plan, seed=9)
        dec = ref_decoder(cfg)
        load_params(dec, params)
        dec.train()
        x, sw = O.synth_decoder_inputs(name, batch=2, seed=9)
        with torch.enable_grad():
            x = [t.clone().requires_grad_(True) for t in x]
            sw = [t.clone().requires_grad_(True) for t in sw] if isinstance(sw, list) else sw.clone().requires_grad_(True)
            y = dec(x, sw)
            r = torch.randn(y.shape, generator=torch.Generator().manual_seed(10))
            (y * r).sum().backward()
        arrs = {f'x{i}': t.detach() for i, t in enumerate(x)}
        arrs.update({f'gx{i}': (t.grad if t.grad is not None else torch.zeros_like(t)) for i, t in enumerate(x)})
        if isinstance(sw, list):
            arrs.update({f'w{i}': t.detach() for i, t in enumerate(sw)})
            arrs.update({f'gw{i}': t.grad for i, t in enumerate(sw)})
        else:
            arrs['s'] = sw.detach()
            arrs['gs'] = sw.grad
        arrs.update({'p.' + k: v for k, v in params.items()})
        arrs.update({'g.' + k: v.grad for k, v in dec.named_parameters() if v.grad is not None})
        arrs.update({'after.' + k: v for k, v in dec.state_dict().items() if 'running_' in k})
        arrs['y'] = y.detach()
        arrs['r'] = r
        save('train_' + name, **arrs)
        print(name, 'train y absmax', float(y.abs().max()), 'n grads', sum(1 for k in arrs if k.startswith('g.')))


def gen_train_step():
    """TWO optimisation steps of the tiny v1_0 decoder with the REFERENCE's loss / optimiser / LR policy
    (BootstrappedCrossEntropyLoss(k, thresh, ignore_index=255), Adam(lr=1e-3, betas=(0.5, 0.999)), PolyLR per batch:
    train.py:118-136, 240-262): losses, learning rates and every parameter / BN buffer after the steps."""
    from hyperseg.losses.bootstrapped_ce_loss import BootstrappedCrossEntropyLoss
    from hyperseg.utils.polylr import PolyLR
    name = 't_v1_0'
    cfg = TINY[name]
    O.CONFIGS[name] = cfg
    plan = O.config_plan(name)
    params = O.synth_decoder_params(plan, seed=21)
    dec = ref_decoder(cfg)
    load_params(dec, params)
    dec.train()
    x, s = O.synth_decoder_inputs(name, batch=2, seed=21)
    g = torch.Generator().manual_seed(22)
    with torch.no_grad():
        y0 = dec(x, s)
    target = torch.randint(0, y0.shape[1], (y0.shape[0],) + tuple(y0.shape[2:]), generator=g)
    target[torch.rand(target.shape, generator=g) < 0.1] = 255
    k = y0.shape[2] * y0.shape[3] // 8
    crit = BootstrappedCrossEntropyLoss(k=k, thresh=0.3, ignore_index=255)
    start = {kk: v.clone() for kk, v in dec.state_dict().items()}
    opt = torch.optim.Adam(dec.parameters(), lr=1e-3, betas=(0.5, 0.999))
    sched = PolyLR(opt, 10, 0.9)
    losses, lrs = [], []
    with torch.enable_grad():
        for it in range(2):
            pred = dec(x, s)
            loss = crit(pred, target)
            opt.zero_grad()
            loss.backward()
            opt.step()
            sched.step()
            losses.append(float(loss))
            lrs.append(opt.param_groups[0]['lr'])
            if it == 0:
                pred0 = pred.detach().clone()
    arrs = {f'x{i}': t for i, t in enumerate(x)}
    arrs['s'] = s
    arrs['target'] = target
    arrs['k'] = np.array(k)
    arrs['pred0'] = pred0
    arrs['losses'] = np.array(losses, dtype=np.float64)
    # the other branch of the bootstrap rule (the (k+1)-th loss is below the threshold -> plain top-k)
    arrs['loss_topk'] = np.array(float(BootstrappedCrossEntropyLoss(k=k, thresh=5.0, ignore_index=255)(pred0, target)))
    arrs['loss_thresh'] = np.array(float(BootstrappedCrossEntropyLoss(k=k, thresh=0.3, ignore_index=255)(pred0, target)))
    arrs['lrs'] = np.array(lrs, dtype=np.float64)
    arrs.update({'start.' + kk: v for kk, v in start.items() if 'num_batches' not in kk})
    arrs.update({'end.' + kk: v for kk, v in dec.state_dict().items() if 'num_batches' not in kk})
    save('train_step_t_v1_0', **arrs)
    print('train_step losses', losses, 'lrs', lrs, 'k', k)


def gen_confusion_matrix():
    """The reference's ConfusionMatrix (hyperseg/utils/seg_utils.py:5-36) on seeded label maps incl. ignored targets."""
    from hyperseg.utils.seg_utils import ConfusionMatrix
    g = torch.Generator().manual_seed(31)
    n = 7
    cm = ConfusionMatrix(n)
    ts, ps = [], []
    for _ in range(3):
        t = torch.randint(0, n, (2, 9, 11), generator=g)
        t[torch.rand(t.shape, generator=g) < 0.15] = 255
        p = torch.randint(0, n - 1, (2, 9, 11), generator=g)          # class n-1 never predicted: a zero column
        cm.update(t.flatten(), p.flatten())
        ts.append(t)
        ps.append(p)
    acc_global, acc, iu = cm.compute()
    save('confusion_matrix', target=torch.stack(ts), pred=torch.stack(ps), mat=cm.mat, acc_global=acc_global, acc=acc, iu=iu)


# ------------------------------------------------------------------------------ whole models (boundary)
MODEL_KW = {
    'M': dict(mod='v1_0', name='efficientnet-b1', num_classes=19, kw=dict(
        levels=2, out_feat_scale=[1., .25, .25, .25, .25], kernel_sizes=[1, 1, 1, 3, 3],
        level_channels=[64, 32, 16, 16, 16], expand_ratio=2, with_out_fc=False, decoder_dropout=None,
        weight_groups=[32, 16, 8, 16, 4], decoder_groups=1, inference_hflip=True,
        coords_res=[(512, 512), (512, 1024)])),
    'S': dict(mod='unify', name='efficientnet-b1', num_classes=19, kw=dict(
        levels=2, out_feat_scale=[1., .166, .2, .25, .4], kernel_sizes=[1, 1, 1, 3, 3],
        level_channels=[32, 16, 8, 8, 8], expand_ratio=2, with_out_fc=False, decoder_dropout=None,
        weight_groups=[32, 16, 8, 16, 4], decoder_groups=1, inference_hflip=True, unify_level=4,
        coords_res=[(768, 768), (768, 1536)])),
    'L': dict(mod='v0_1', name='efficientnet-b3', num_classes=21, kw=dict(
        levels=3, kernel_sizes=(1, 1, 3, 3, 3, 3), expand_ratio=2, inference_hflip=True, with_out_fc=False,
        decoder_dropout=None, weight_groups=16)),
    # CamVid HyperSeg-L (configs/train/camvid_efficientnet_b1_hyperseg-l.py:35-38): the six-level v1_0 model
    'Lc': dict(mod='v1_0', name='efficientnet-b1', num_classes=12, kw=dict(
        levels=2, kernel_sizes=(1, 1, 1, 3, 3, 3), level_channels=[64, 32, 16, 16, 16, 16], expand_ratio=2,
        inference_hflip=True, with_out_fc=False, decoder_dropout=None, weight_groups=[64, 32, 32, 16, 8, 8],
        coords_res=[(768, 768), (768, 1024)])),
}


def gen_models():
    """HyperGen end to end with name-keyed weights (tests/util_weights.py): pins the stock-PyTorch encoder,
    the context head and the factory kwargs of the build to the reference."""
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    from util_weights import fill_by_name
    mods = {'v1_0': v1, 'unify': vu, 'v0_1': v0}
    for tag, spec in MODEL_KW.items():
        kw = {k: (list(v) if isinstance(v, list) else v) for k, v in spec['kw'].items()}
        model = mods[spec['mod']].hyperseg_efficientnet(spec['name'], False, num_classes=spec['num_classes'], **kw)
        fill_by_name(model.eval(), seed=11)
        x = torch.rand(1, 3, 128, 256, generator=torch.Generator().manual_seed(12))
        feats = model.backbone(x)
        sig = model.weight_mapper(feats[-1])
        if isinstance(sig, (list, tuple)):          # v0_1: list of per-level weight tensors
            sig = torch.cat([t[:, ::97] for t in sig], dim=1)
        y = model(x)
        keys = [k for k in model.state_dict() if 'num_batches' not in k]
        save(f'model_{tag}', x=x, y=y[:, :, 1::5, 2::7].contiguous(), y_absmax=y.abs().max(),
             y_shape=np.array(y.shape), signal=sig[:, ::13].contiguous(),
             feat_absmax=np.array([float(f.abs().max()) for f in feats]),
             n_keys=len(keys), key_hash=np.array([hash_str(' '.join(keys))]),
             shape_hash=np.array([hash_str(' '.join(str(tuple(model.state_dict()[k].shape)) for k in keys))]),
             mask=y.argmax(1)[:, 1::5, 2::7].to(torch.uint8),
             margin=(y.topk(2, dim=1).values[:, 0] - y.topk(2, dim=1).values[:, 1])[:, 1::5, 2::7].contiguous())
        print(tag, tuple(y.shape), 'absmax', float(y.abs().max()), 'signal absmax', float(sig.abs().max()))


def hash_str(s):
    import hashlib
    return int.from_bytes(hashlib.sha256(s.encode()).digest()[:7], 'little')


def gen_model_pyramid():
    """HyperGen's list-input inference mode in the REFERENCE (hyperseg_v1_0.py:70-91): an image pyramid of two scales with
    horizontal-flip test-time augmentation (max over the flip pair, mean over the scales, coarse scale resized to the first)."""
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    from util_weights import fill_by_name
    spec = MODEL_KW['M']
    kw = {k: (list(v) if isinstance(v, list) else v) for k, v in spec['kw'].items()}
    model = v1.hyperseg_efficientnet(spec['name'], False, num_classes=spec['num_classes'], **kw)
    fill_by_name(model.eval(), seed=11)
    assert model.inference_hflip and model.inference_gather == 'mean'
    g = torch.Generator().manual_seed(13)
    x0 = torch.rand(1, 3, 128, 256, generator=g)
    x1 = torch.rand(1, 3, 64, 128, generator=g)
    y = model([x0, x1])
    top2 = y.topk(2, dim=1).values
    save('model_M_pyramid', x0=x0, x1=x1, y=y[:, :, 1::3, 2::5].contiguous(), y_absmax=y.abs().max(), y_shape=np.array(y.shape),
         mask=y.argmax(1)[:, 1::3, 2::5].to(torch.uint8), margin=(top2[:, 0] - top2[:, 1])[:, 1::3, 2::5].contiguous())


def gen_model_full():
    """HyperSeg-M at the BENCHED size (1024x512, bs 1) in the REFERENCE: the fixture the benched configuration
    (prepared encoder + split GEMM + HIP decoder + graph replay) is held to directly (VERDICT r3 missing #6).  The image
    is regenerated from its seed by the test (torch's CPU generator); a checksum + a sample pin that it is the same image.
    Stored: a strided logits sample, the FULL argmax mask, and the set of pixels whose top-2 margin is clear (bit-packed)."""
    sys.path.insert(0, os.path.join(REPO, 'tests'))
    from util_weights import fill_by_name
    spec = MODEL_KW['M']
    kw = {k: (list(v) if isinstance(v, list) else v) for k, v in spec['kw'].items()}
    model = v1.hyperseg_efficientnet(spec['name'], False, num_classes=spec['num_classes'], **kw)
    fill_by_name(model.eval(), seed=11)
    x = torch.rand(1, 3, 512, 1024, generator=torch.Generator().manual_seed(14))
    y = model(x)
    top2 = y.topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    absmax = float(y.abs().max())
    clear = (margin > 1e-3 * absmax).numpy()
    save('model_M_full', seed=np.array(14), x_sum=x.double().sum(), x_sample=x[:, :, 7::61, 11::67].contiguous(),
         y=y[:, :, 3::16, 5::16].contiguous(), y_absmax=y.abs().max(), y_shape=np.array(y.shape),
         mask=y.argmax(1).to(torch.uint8), clear_bits=np.packbits(clear), clear_thresh=np.array(1e-3 * absmax),
         n_clear=np.array(int(clear.sum())), margin_sample=margin[:, 3::16, 5::16].contiguous())
    print('M full', tuple(y.shape), 'absmax', absmax, 'clear', int(clear.sum()), 'of', clear.size)


def gen_checkpoint():
    """Row f4: the REFERENCE's own ``get_arch`` / ``save_checkpoint`` (hyperseg/utils/utils.py:61-144) outputs.
    (1) arch strings of the config files' model partials exactly as train.py:203 forms them
    (``get_arch(model, num_classes=len(classes))``; kwargs verbatim from configs/train/*.py, ``pretrained`` as the
    config passes it -- by keyword) and of the configs' optimizer / scheduler partials, plus the string / nested / non-object
    cases of utils.py:112-143;  (2) a checkpoint FILE written by the reference's ``save_checkpoint`` with the dict
    train.py:267-274 stores (DataParallel-prefixed keys in, Adam + PolyLR state), for a tiny v1_0 decoder whose arch
    string the reference's ``get_arch`` produced: ``tests/golden/ref_ckpt_latest.pth`` / ``_best.pth`` (data: tensors,
    strings, numbers)."""
    import shutil
    import tempfile
    from functools import partial
    import hyperseg.utils.utils as U
    from hyperseg.utils.polylr import PolyLR
    from hyperseg.utils.obj_factory import obj_factory as ref_factory
    cfgs = {
        # configs/train/cityscapes_efficientnet_b1_hyperseg-m.py:36-40
        'M': (v1.hyperseg_efficientnet, ('efficientnet-b1',), dict(
            pretrained=True, levels=2, out_feat_scale=[1., 0.25, 0.25, 0.25, 0.25], kernel_sizes=[1, 1, 1, 3, 3],
            level_channels=[64, 32, 16, 16, 16], expand_ratio=2, with_out_fc=False, decoder_dropout=None,
            weight_groups=[32, 16, 8, 16, 4], decoder_groups=1, inference_hflip=True,
            coords_res=[(512, 512), (512, 1024)]), 19),
        # configs/train/cityscapes_efficientnet_b1_hyperseg-s.py:36-40
        'S': (vu.hyperseg_efficientnet, ('efficientnet-b1',), dict(
            pretrained=True, levels=2, out_feat_scale=[1., 0.166, 0.2, 0.25, 0.4], kernel_sizes=[1, 1, 1, 3, 3],
            level_channels=[32, 16, 8, 8, 8], expand_ratio=2, with_out_fc=False, decoder_dropout=None,
            weight_groups=[32, 16, 8, 16, 4], decoder_groups=1, inference_hflip=True, unify_level=4,
            coords_res=[(768, 768), (768, 1536)]), 19),
        # configs/train/camvid_efficientnet_b1_hyperseg-s.py:35-38
        'Sc': (v1.hyperseg_efficientnet, ('efficientnet-b1',), dict(
            pretrained=True, levels=2, kernel_sizes=(1, 1, 1, 3, 3), level_channels=[64, 32, 16, 16, 16],
            expand_ratio=2, with_out_fc=False, decoder_dropout=None, weight_groups=[64, 32, 32, 16, 8],
            decoder_groups=1, inference_hflip=True, coords_res=[(576, 576), (576, 768)]), 12),
        # configs/train/vocsbd_efficientnet_b3_hyperseg-l.py:32-34
        'L': (v0.hyperseg_efficientnet, ('efficientnet-b3',), dict(
            pretrained=True, levels=3, kernel_sizes=(1, 1, 3, 3, 3, 3), expand_ratio=2, inference_hflip=True,
            with_out_fc=False, decoder_dropout=None, weight_groups=16), 21),
    }
    arrs = {}
    for tag, (fn, args, kw, ncls) in cfgs.items():
        arrs[f'arch.{tag}'] = U.get_arch(partial(fn, *args, **kw), num_classes=ncls)
        arrs[f'classes.{tag}'] = np.array(ncls)
    arrs['arch.adam'] = U.get_arch(partial(torch.optim.Adam, lr=1e-3, betas=(0.5, 0.999)))
    arrs['arch.polylr'] = U.get_arch(partial(PolyLR, power=0.9, max_epoch=90000))
    # a string WITH arguments: utils.py:116 evals 'extract_args(...)', a name utils.py never imports (it lives in
    # obj_factory.py:31) -> the reference raises NameError here.  Recorded as a fact; the build's get_arch handles the case.
    try:
        U.get_arch("hyperseg.models.hyperseg_v1_0.hyperseg_efficientnet('efficientnet-b1', levels=2)", num_classes=3)
        arrs['arch.str_args_raises'] = ''
    except Exception as e:                                      # noqa: BLE001
        arrs['arch.str_args_raises'] = type(e).__name__
    try:                                                        # utils.py:126: [] + () -> TypeError for any plain string
        arrs['arch.str_plain_raises'] = ''
        U.get_arch('torch.nn.ReLU')
    except Exception as e:                                      # noqa: BLE001
        arrs['arch.str_plain_raises'] = type(e).__name__
    arrs['arch.nested'] = U.get_arch(partial(max, partial(min, 1)))
    arrs['arch.not_eval'] = U.get_arch(partial(torch.nn.ReLU6, True), eval_partial=False)
    arrs['arch.none_is_none'] = np.array(U.get_arch(42) is None)

    # (2) a checkpoint file written by the reference
    cfg = TINY['t_v1_0']
    dec_partial = partial(v1.MultiScaleDecoder, cfg['feat'], cfg['signal'], cfg['num_classes'], cfg['kernel_sizes'], 1,
                          cfg['level_channels'], expand_ratio=cfg['expand_ratio'], weight_groups=list(cfg['weight_groups']))
    arch = U.get_arch(dec_partial)
    dec = ref_factory(arch)
    O.CONFIGS['t_v1_0'] = cfg
    load_params(dec, O.synth_decoder_params(O.config_plan('t_v1_0'), seed=33))
    opt = torch.optim.Adam(dec.parameters(), lr=1e-3, betas=(0.5, 0.999))
    sched = PolyLR(opt, 10, 0.9)
    x, s = O.synth_decoder_inputs('t_v1_0', batch=1, seed=33)
    with torch.enable_grad():
        dec.train()
        dec(x, s).square().mean().backward()
        opt.step()
        sched.step()
    dec.eval()
    y = dec(x, s)
    wrapped = torch.nn.DataParallel(dec)                      # train.py:207 wraps the model: keys get 'module.'
    tmp = tempfile.mkdtemp()
    U.save_checkpoint(tmp, 'model', {'epoch': 4, 'state_dict': wrapped.state_dict(), 'optimizer': opt.state_dict(),
                                     'scheduler': sched.state_dict(), 'best_iou': 0.625, 'arch': arch}, True)
    for suffix in ('latest', 'best'):
        shutil.copyfile(os.path.join(tmp, f'model_{suffix}.pth'), os.path.join(HERE, f'ref_ckpt_{suffix}.pth'))
    ck = torch.load(os.path.join(tmp, 'model_latest.pth'), weights_only=True)
    arrs['ckpt.files'] = np.array(sorted(os.listdir(tmp)))
    arrs['ckpt.top_keys'] = np.array(list(ck.keys()))
    arrs['ckpt.state_keys'] = np.array(list(ck['state_dict'].keys()))
    arrs['ckpt.arch'] = arch
    arrs['ckpt.y'] = y
    arrs.update({f'ckpt.x{i}': t for i, t in enumerate(x)})
    arrs['ckpt.s'] = s
    shutil.rmtree(tmp)
    save('checkpoint_ref', **arrs)
    for k in sorted(arrs):
        if k.startswith('arch.'):
            print(k, arrs[k])


if __name__ == '__main__':
    ALL = [gen_meta_conv, gen_meta_conv_general, gen_meta_patch, gen_meta_sequential, gen_hyper_patch, gen_ir_v1, gen_ir_v0, gen_divide_feature,
           gen_decoders, gen_train, gen_train_step, gen_confusion_matrix, gen_models, gen_model_pyramid, gen_model_full, gen_checkpoint]
    only = set(sys.argv[1:])            # e.g. "python make_golden.py gen_train_step" regenerates one fixture family
    for fn in ALL:
        if not only or fn.__name__ in only:
            fn()
