"""Pins the CPU oracle (oracle/hyperseg_oracle.py) to fixtures produced by the reference itself
(tests/golden/make_golden.py).  Runs on CPU; no GPU, no reference needed."""
import numpy as np
import pytest
import torch

from conftest import G, bn_of, rel_err, sub
from oracle import hyperseg_oracle as O

TOL = 2e-6   # the oracle re-associates fp32 sums; observed <= 1.5e-6 relative


def test_known_answer(golden):
    # the reference's only known-answer test: meta_conv.py:233-242 prints tensor(9.)
    g = golden('meta_conv_known_answer')
    assert float(g['out_max']) == 9.0
    x = torch.ones(4, 3, 64, 64)
    x[0::2] = 0
    w = torch.ones(4, 27)
    w[0::2] = 0
    assert float(O.meta_conv2d(x, w, 3, 3, padding=1, groups=3).max()) == 9.0


def test_meta_conv2d(golden):
    g = golden('meta_conv2d')
    for i in range(int(g['n'])):
        cin, cout, k, pad, groups = [int(v) for v in g[f'{i}.cfg']]
        y = O.meta_conv2d(g[f'{i}.x'], g[f'{i}.w'], cout, k, padding=pad, groups=groups,
                          padding_mode=str(g[f'{i}.mode']))
        assert rel_err(y, g[f'{i}.y']) < TOL, i


def test_meta_conv2d_general(golden):
    """MetaConv2d with non-square kernels, stride, dilation, any padding (meta_conv.py:141-186): reference-made outputs."""
    g = golden('meta_conv2d_general')
    for i in range(int(g['n'])):
        cin, cout, kh, kw, sh, sw, ph, pw, dh, dw, groups = [int(v) for v in g[f'{i}.cfg']]
        y = O.meta_conv2d(g[f'{i}.x'], g[f'{i}.w'], cout, (kh, kw), stride=(sh, sw), padding=(ph, pw), dilation=(dh, dw),
                          groups=groups, padding_mode=str(g[f'{i}.mode']))
        assert tuple(y.shape) == tuple(g[f'{i}.y'].shape)
        assert rel_err(y, g[f'{i}.y']) < 1e-5, i


def test_meta_patch_conv2d(golden):
    g = golden('meta_patch_conv2d')
    for i in range(int(g['n'])):
        cin, cout, k, groups = [int(v) for v in g[f'{i}.cfg']]
        y = O.meta_patch_conv2d(g[f'{i}.x'], g[f'{i}.w'], cout, k, padding=k // 2, groups=groups)
        assert y.shape == g[f'{i}.y'].shape
        assert rel_err(y, g[f'{i}.y']) < TOL, i
    p = sub(g, 'blk.p.')
    y = O.act(O.bn_eval(O.patch_conv_k1(g['blk.x'], g['blk.w'], 5), bn_of(p, '1')), O.ACT_RELU)
    assert rel_err(y, g['blk.y']) < TOL


def test_meta_sequential_slicing(golden):
    g = golden('meta_sequential')
    hp0, hp1 = [int(v) for v in g['hp']]
    assert list(g['ranges']) == [0, hp0, hp0, hp0 + hp1]

    def run(w0, w1):
        y = O.patch_conv_k1(g['x'], w0, 6).clamp(min=0)
        return O.patch_conv_kxk(y, w1, 3, 3, 1)
    y = run(g['w'][:, :hp0], g['w'][:, hp0:hp0 + hp1])
    for key in ('y_tensor', 'y_list', 'y_long'):
        assert rel_err(y, g[key]) < TOL, key


def test_hyper_patch_v1(golden):
    g = golden('hyper_patch_v1')
    cin, cout, cs, idx, grp, hp = [int(v) for v in g['np.cfg']]
    assert hp == 35 and g['np.w_s2w'].shape[0] == 36          # next_multiply padding
    wt = O.signal2weights(g['np.s'], g['np.w_s2w'], idx, cs, grp, hp)
    assert rel_err(wt, g['np.wt']) < TOL
    assert rel_err(O.patch_conv_k1(g['np.x'], wt, cout), g['np.y']) < TOL
    cin, cout, cs, idx, grp, hp = [int(v) for v in g['pc.cfg']]
    wt = O.signal2weights(g['pc.s'], g['pc.w_s2w'], idx, cs, grp, hp)
    assert rel_err(O.patch_conv_kxk(g['pc.x'], wt, cout, 3, 1), g['pc.y']) < TOL
    cin, cout, cs, idx, grp, hp = [int(v) for v in g['blk.cfg']]
    p = sub(g, 'blk.p.')
    # MetaSequential hands s[:, 0:hp] (clamped to the 20 available channels) to the module
    wt = O.signal2weights(g['blk.s'][:, :hp], p['0.signal2weights.weight'], idx, cs, grp, hp)
    y = O.act(O.bn_eval(O.patch_conv_k1(g['blk.x'], wt, cout), bn_of(p, '1')), O.ACT_RELU)
    assert rel_err(y, g['blk.y']) < TOL


def test_inverted_residual_v1(golden):
    g = golden('inverted_residual_v1')
    for i in range(int(g['n'])):
        cin, cout, hid, cs, idx, grp, hp = [int(v) for v in g[f'{i}.cfg']]
        p = sub(g, f'{i}.p.')
        wt = O.signal2weights(g[f'{i}.s'], p['signal2weights.weight'], idx, cs, grp, hp)
        assert rel_err(wt, g[f'{i}.wt']) < TOL
        y = O.patch_inverted_residual_v1(g[f'{i}.x'], wt, hid, cout, bn_of(p, 'bn1'), bn_of(p, 'bn2'),
                                         bn_of(p, 'bn3'))
        assert rel_err(y, g[f'{i}.y']) < TOL, i
    cin, cout, hid = [int(v) for v in g['u.cfg']]
    p = sub(g, 'u.p.')
    y = O.patch_inverted_residual_v1(g['u.x'], g['u.wt'], hid, cout, bn_of(p, 'bn1'), bn_of(p, 'bn2'),
                                     bn_of(p, 'bn3'))
    assert rel_err(y, g['u.y']) < TOL


def test_inverted_residual_v0(golden):
    g = golden('inverted_residual_v0')
    for i in range(int(g['n'])):
        cin, cout, hid = [int(v) for v in g[f'{i}.cfg']]
        p = sub(g, f'{i}.p.')
        y = O.patch_inverted_residual_v0(g[f'{i}.x'], g[f'{i}.wt'], hid, cout, bn_of(p, 'conv.0.1'),
                                         bn_of(p, 'conv.1.1'), bn_of(p, 'conv.2.1'))
        assert rel_err(y, g[f'{i}.y']) < TOL, i


def test_divide_feature(golden):
    g = golden('divide_feature')
    expect = {'M': [416, 224, 128, 192, 320], 'S': [576, 128, 64, 512], 'Sc': [448, 256, 256, 192, 128]}
    for name, want in expect.items():
        got = O.divide_feature(1280, [int(v) for v in g[f'{name}.targets']], int(g[f'{name}.min_unit']))
        assert list(got) == want == [int(v) for v in g[f'{name}.split']]
    for i in range(12):
        t = [int(v) for v in g[f'r{i}.targets']]
        mu, total = int(g[f'r{i}.min_unit']), int(g[f'r{i}.total'])
        assert list(O.divide_feature(total, t, mu)) == [int(v) for v in g[f'r{i}.split']], i
        assert list(O.divide_feature_legacy(total, t, mu)) == [int(v) for v in g[f'r{i}.legacy']], i


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


@pytest.mark.parametrize('name', list(TINY))
def test_tiny_decoders(golden, name):
    g = golden('decoder_' + name)
    c = TINY[name]
    plan = O.decoder_plan(c['variant'], c['feat'], c['signal'], c['num_classes'], c['kernel_sizes'],
                          c.get('level_channels'), c['expand_ratio'], c.get('weight_groups', 1),
                          c.get('unify_level'))
    params = sub(g, 'p.')
    x = [g[f'x{i}'] for i in range(6)]
    if c['variant'] == 'v0_1':
        y = O.decoder_v0_1(plan, params, x, [g[f'w{i}'] for i in range(6)])
    else:
        assert [sw['hp'] for sw in plan['s2w']] == [int(v) for v in g['hyper_params']]
        fn = O.decoder_v1_0 if c['variant'] == 'v1_0' else O.decoder_unify
        y = fn(plan, params, x, g['s'])
    assert y.shape == g['y'].shape
    assert rel_err(y, g['y']) < TOL
    assert bool((y.argmax(1) == g['y'].argmax(1)).all())


@pytest.mark.parametrize('name', ['M', 'S', 'Sc', 'L', 'Lc'])
def test_full_config_samples(golden, name):
    """Full BASELINE shapes: strided logits sample and margin-aware mask agreement vs the reference."""
    g = golden('decoder_full_configs')
    assert int(g[f'{name}.oracle_flips_margin_gt_1e-4']) == 0
    y = O.run_config(name, batch=int(g[f'{name}.batch']), seed=0)
    ys = y[:, :, 3::37, 5::41]
    ref = g[f'{name}.logits_sample']
    assert float((ys - ref).abs().max()) < 1e-5 * float(g[f'{name}.logits_absmax'])
    ok = g[f'{name}.margin_sample'] > 1e-4
    assert bool((ys.argmax(1).to(torch.uint8)[ok] == g[f'{name}.mask_sample'][ok]).all())


def test_upsample_and_coords_match_torch():
    g = torch.Generator().manual_seed(0)
    p = torch.randn(2, 3, 5, 7, generator=g)
    for size in [(10, 14), (15, 9), (5, 7)]:
        want = torch.nn.functional.interpolate(p, size, mode='bilinear', align_corners=False)
        assert torch.allclose(O.upsample_bilinear(p, size), want, atol=1e-6)
    c = O.image_coords(4, 6)
    assert torch.equal(c[0, 0], torch.linspace(-1, 1, 6)) and torch.equal(c[1, :, 0], torch.linspace(-1, 1, 4))
    x = torch.randn(1, 2, 5, 6, generator=g)
    for mode in ('reflect', 'replicate', 'circular'):
        assert torch.equal(O.pad2d(x, (1, 1), mode), torch.nn.functional.pad(x, (1, 1, 1, 1), mode=mode))


def test_cpu_port_matches_oracle():
    """oracle/cpu_port.py (the timed CPU baseline of bench.py) computes what the oracle computes."""
    from oracle import cpu_port as P
    for name, size in (('M', (128, 256)), ('Sc', (96, 64))):
        plan = O.config_plan(name)
        params = O.synth_decoder_params(plan, seed=3)
        x, s = O.synth_decoder_inputs(name, batch=2, seed=3, size=size)
        assert rel_err(P.decoder_v1_0(plan, params, x, s), O.decoder_v1_0(plan, params, x, s)) < TOL


def test_train_step_oracle_vs_reference(golden):
    """Train-mode forward + autograd of the oracle == the reference's own train-mode forward, gradients and updated
    BatchNorm running statistics (fixture train_t_v1_0: loss = sum(y * r))."""
    g = golden('train_t_v1_0')
    c = TINY['t_v1_0']
    plan = O.decoder_plan(c['variant'], c['feat'], c['signal'], c['num_classes'], c['kernel_sizes'],
                          c['level_channels'], c['expand_ratio'], c['weight_groups'])
    params = {k: v.clone().requires_grad_(v.dtype.is_floating_point and 'running' not in k) for k, v in sub(g, 'p.').items()}
    x = [g[f'x{i}'].clone().requires_grad_(True) for i in range(6)]
    s = g['s'].clone().requires_grad_(True)
    y, stats = O.decoder_v1_0(plan, params, x, s, training=True)
    assert rel_err(y.detach(), g['y']) < 1e-5
    (y * g['r']).sum().backward()
    for i in range(1, 6):                       # x[0] (the image) only sets the output size in a 5-level decoder
        assert rel_err(x[i].grad, g[f'gx{i}']) < 1e-4, i
    assert x[0].grad is None and float(g['gx0'].abs().max()) == 0.0
    assert rel_err(s.grad, g['gs']) < 1e-4
    for k, v in sub(g, 'g.').items():
        assert rel_err(params[k].grad, v) < 1e-4, k
    for k, v in sub(g, 'after.').items():
        assert rel_err(stats[k], v) < 1e-5, k


def test_training_harness_loss_and_lr_vs_reference(golden):
    """hyperseg_amd.training (bootstrapped CE, PolyLR) against values captured from the reference's own classes
    (make_golden.gen_train_step): both branches of the bootstrap rule, ignore_index pixels, the per-batch LR decay."""
    from hyperseg_amd.training import BootstrappedCrossEntropyLoss, PolyLR, bootstrapped_cross_entropy
    g = golden('train_step_t_v1_0')
    k = int(g['k'])
    pred, target = g['pred0'], g['target']
    assert bool((target == 255).any())
    a = float(bootstrapped_cross_entropy(pred, target, k, 0.3, ignore_index=255))
    b = float(BootstrappedCrossEntropyLoss(k=k, thresh=5.0, ignore_index=255)(pred, target))
    assert abs(a - float(g['loss_thresh'])) < 1e-6 * abs(a) + 1e-7
    assert abs(b - float(g['loss_topk'])) < 1e-6 * abs(b) + 1e-7
    assert abs(a - b) > 1e-3                                   # the two branches really differ on this input
    assert abs(a - float(g['losses'][0])) < 1e-6
    # the device-resident form of the rule (what a CUDA tensor gets) == the reference's statement, values and gradients, both branches
    from hyperseg_amd.training import bootstrap_mean_on_device, bootstrap_mean_reference
    gen = torch.Generator().manual_seed(3)
    for th in (0.3, 2.5, 5.0):
        v = (torch.rand(5000, generator=gen) * 6).requires_grad_()
        r0, r1 = bootstrap_mean_reference(v, 1000, th), bootstrap_mean_on_device(v, 1000, th)
        g0, g1 = torch.autograd.grad(r0, v)[0], torch.autograd.grad(r1, v)[0]
        assert abs(float(r0) - float(r1)) < 1e-6 * abs(float(r0)) and torch.allclose(g0, g1, rtol=1e-6, atol=0)
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=1e-3, betas=(0.5, 0.999))
    sched = PolyLR(opt, 10, 0.9)
    lrs = []
    for _ in range(2):
        opt.step()
        sched.step()
        lrs.append(opt.param_groups[0]['lr'])
    assert np.allclose(lrs, g['lrs'].numpy() if hasattr(g['lrs'], 'numpy') else g['lrs'], rtol=1e-12)


def test_fps_harness_confusion_matrix_and_plumbing(golden):
    """hyperseg_amd.fps: ConfusionMatrix against the reference's own class (fixture confusion_matrix.npz, incl. ignored
    targets and a never-predicted class), remove_bn, and the per-iteration timing loop on a CPU-only box (guarded sync)."""
    from hyperseg_amd.fps import ConfusionMatrix, measure_fps, remove_bn
    g = golden('confusion_matrix')
    cm = ConfusionMatrix(int(g['mat'].shape[0]))
    for t, p in zip(g['target'], g['pred']):
        cm.update(t.flatten(), p.flatten())
    assert torch.equal(cm.mat, g['mat'])
    acc_global, acc, iu = cm.compute()
    assert abs(float(acc_global) - float(g['acc_global'])) < 1e-7
    assert torch.allclose(acc, g['acc'], rtol=0, atol=1e-7) and torch.allclose(iu, g['iu'], rtol=0, atol=1e-7)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 5, 1), torch.nn.BatchNorm2d(5), torch.nn.Sequential(torch.nn.BatchNorm2d(5)))
    remove_bn(net)
    assert not any(isinstance(m, torch.nn.BatchNorm2d) for m in net.modules())
    gx, gt = G(1032), G(1033)
    batches = [(torch.rand(2, 3, 4, 6, generator=gx), torch.randint(0, 5, (2, 4, 6), generator=gt)) for _ in range(3)]
    res = measure_fps(net.eval(), batches, torch.device('cpu'), 5)
    assert res['frames'] == 6 and res['pass'] == 1 and res['fps'] > 0 and 0.0 <= res['mean_iou'] <= 1.0
