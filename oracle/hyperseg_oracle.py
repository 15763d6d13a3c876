"""CPU oracle for the HyperSeg decoder hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-torch (CPU, dtype-generic) restatement of the arithmetic of the
reference decoder.  It is the *checker* used by
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.
Nothing under ``hyperseg_amd/`` may import it: the product path is the HIP library and
fails loudly without it.

Parity status: PINNED.  Every function below is checked in ``tests/test_oracle_golden.py``
against fixtures in ``tests/golden/*.npz`` that were produced by importing the reference
itself (``tests/golden/make_golden.py``, run in the build container where
``/root/reference`` exists).

Style: no unfold / fold / "fold batch into groups" tricks.  Patches are addressed by
explicit reshapes of the (fh, ph, fw, pw) pixel grid or explicit index gathers, and the
contractions are einsums, so that every index in Appendix A is visible.

Reference anchors (file:line under /root/reference):
  signal2weights ............ hyperseg/models/hyperseg_v1_0.py:473-484, 315-326
  Op A  k=1 patch conv ...... hyperseg/models/hyperseg_v1_0.py:486-498
  Op B  kxk patch conv ...... hyperseg/models/layers/meta_patch.py:35-57 + meta_conv.py:163-186
  Op C  fused inv. residual . hyperseg/models/hyperseg_v1_0.py:328-376
  Op D  v0_1 inv. residual .. hyperseg/models/hyperseg_v0_1.py:205-237
  Op E  stage glue .......... hyperseg/models/hyperseg_v1_0.py:203-253
  MetaConv2d ................ hyperseg/models/layers/meta_conv.py:163-186
  MetaSequential slicing .... hyperseg/models/layers/meta_sequential.py:19-40
  divide_feature ............ hyperseg/models/hyperseg_v1_0.py:763-810
  divide_feature_legacy ..... hyperseg/models/hyperseg_v0_1.py:366-406
"""
import math
from itertools import groupby

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5          # nn.BatchNorm2d default, never overridden by the reference
BN_MOMENTUM = 0.1

ACT_NONE, ACT_RELU, ACT_RELU6 = 0, 1, 2


# --------------------------------------------------------------------------------------
# small pieces
# --------------------------------------------------------------------------------------
def next_multiply(x, base):
    """Smallest multiple of ``base`` that is >= x (hyperseg_v1_0.py:451-452)."""
    return int(math.ceil(x / base) * base)


def act(x, kind):
    if kind == ACT_NONE:
        return x
    if kind == ACT_RELU:
        return x.clamp(min=0)
    if kind == ACT_RELU6:
        return x.clamp(min=0, max=6)
    raise ValueError(kind)


def bn_eval(x, bn, ch_dim=1):
    """Inference BatchNorm: per-channel affine from running statistics.

    ``bn`` is a dict with weight, bias, running_mean, running_var (1-D tensors)."""
    shape = [1] * x.dim()
    shape[ch_dim] = -1
    inv = torch.rsqrt(bn['running_var'].to(x.dtype) + BN_EPS)
    scale = bn['weight'].to(x.dtype) * inv
    shift = bn['bias'].to(x.dtype) - bn['running_mean'].to(x.dtype) * scale
    return x * scale.view(shape) + shift.view(shape)


def bn_train(x, bn, ch_dim=1):
    """Training BatchNorm over all dims but ``ch_dim`` (biased var for normalisation,
    unbiased for the running update).  Returns (y, new_running_mean, new_running_var)."""
    dims = [d for d in range(x.dim()) if d != ch_dim]
    n = x.numel() // x.shape[ch_dim]
    mean = x.mean(dim=dims)
    var = x.var(dim=dims, unbiased=False)
    shape = [1] * x.dim()
    shape[ch_dim] = -1
    y = (x - mean.view(shape)) * torch.rsqrt(var.view(shape) + BN_EPS)
    y = y * bn['weight'].view(shape) + bn['bias'].view(shape)
    new_mean = (1 - BN_MOMENTUM) * bn['running_mean'] + BN_MOMENTUM * mean.detach()
    new_var = (1 - BN_MOMENTUM) * bn['running_var'] + BN_MOMENTUM * var.detach() * n / max(n - 1, 1)
    return y, new_mean, new_var


def image_coords(h, w, dtype=torch.float32):
    """(2, h, w): channel 0 = x in [-1, 1] along W, channel 1 = y along H.

    hyperseg_v1_0.py:203-208: linspace endpoints inclusive, meshgrid(y, x)[::-1]."""
    cx = torch.linspace(-1, 1, steps=w, dtype=dtype)
    cy = torch.linspace(-1, 1, steps=h, dtype=dtype)
    return torch.stack([cx.view(1, w).expand(h, w), cy.view(h, 1).expand(h, w)], dim=0)


def _bilinear_taps(out_size, in_size, dtype):
    """Source indices and lambdas of F.interpolate(mode='bilinear', align_corners=False)."""
    scale = in_size / out_size
    dst = torch.arange(out_size, dtype=torch.float64)
    src = (dst + 0.5) * scale - 0.5
    src = src.clamp(min=0)
    i0 = src.floor().to(torch.long).clamp(max=in_size - 1)
    i1 = (i0 + 1).clamp(max=in_size - 1)
    l1 = (src - i0.to(torch.float64)).to(dtype)
    l0 = (1.0 - l1.to(torch.float64)).to(dtype)
    return i0, i1, l0, l1


def upsample_bilinear(p, size):
    """F.interpolate(p, size, mode='bilinear', align_corners=False) as explicit gathers."""
    h_out, w_out = size
    y0, y1, ly0, ly1 = _bilinear_taps(h_out, p.shape[-2], p.dtype)
    x0, x1, lx0, lx1 = _bilinear_taps(w_out, p.shape[-1], p.dtype)
    top = p[..., y0, :]
    bot = p[..., y1, :]
    top = top[..., x0] * lx0 + top[..., x1] * lx1
    bot = bot[..., x0] * lx0 + bot[..., x1] * lx1
    return top * ly0.view(-1, 1) + bot * ly1.view(-1, 1)


def reflect_index(idx, n):
    """Index map of F.pad(mode='reflect'): -1 -> 1, n -> n-2 (no edge repeat)."""
    idx = idx.abs()
    return torch.where(idx >= n, 2 * (n - 1) - idx, idx)


def pad2d(x, pad, mode):
    """Whole-image padding with an explicit index gather; ``pad`` = (py, px) on both sides, or (top, bottom, left, right)."""
    pt, pb, pl, pr = (pad[0], pad[0], pad[1], pad[1]) if len(pad) == 2 else pad
    if pt == pb == pl == pr == 0:
        return x
    h, w = x.shape[-2:]
    if mode == 'zeros':
        out = x.new_zeros(x.shape[:-2] + (h + pt + pb, w + pl + pr))
        out[..., pt:pt + h, pl:pl + w] = x
        return out
    ys = torch.arange(-pt, h + pb)
    xs = torch.arange(-pl, w + pr)
    if mode == 'reflect':
        ys, xs = reflect_index(ys, h), reflect_index(xs, w)
    elif mode == 'replicate':
        ys, xs = ys.clamp(0, h - 1), xs.clamp(0, w - 1)
    elif mode == 'circular':
        ys, xs = ys % h, xs % w
    else:
        raise ValueError(mode)
    return x[..., ys, :][..., xs]


# --------------------------------------------------------------------------------------
# a9: signal2weights (grouped 1x1 conv, bias-free, padded rows truncated)
# --------------------------------------------------------------------------------------
def signal2weights(s, w_s2w, signal_index, signal_channels, groups, hyper_params):
    """Wt = conv1x1_grouped(s[:, idx:idx+Cs]; Wsw)[:, :hp]  ->  (B, hp, fh, fw).

    w_s2w is the nn.Conv2d weight (Wc, Cs/G, 1, 1) with Wc = next_multiply(hp, G).
    Output channel n belongs to group n // (Wc/G) and reads signal channels
    [g*Cs/G, (g+1)*Cs/G) of the slice."""
    b, _, fh, fw = s.shape
    wc = w_s2w.shape[0]
    cs_g = signal_channels // groups
    assert w_s2w.shape[1] == cs_g and wc % groups == 0
    sl = s[:, signal_index:signal_index + signal_channels]
    assert sl.shape[1] == signal_channels, 'signal slice out of range'
    sl = sl.reshape(b, groups, cs_g, fh, fw)
    wg = w_s2w.reshape(groups, wc // groups, cs_g).to(s.dtype)
    out = torch.einsum('gnk,bgkij->bgnij', wg, sl).reshape(b, wc, fh, fw)
    return out[:, :hyper_params]


# --------------------------------------------------------------------------------------
# a2: MetaConv2d (per-sample dynamic conv)
# --------------------------------------------------------------------------------------
def meta_conv2d(x, w, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                padding_mode='zeros'):
    """y[b] = conv2d(pad(x[b]), w[b].view(Cout, Cin/g, kh, kw), groups=g)  (meta_conv.py:163-186).
    Quirk reproduced on purpose: for the non-zero padding modes the reference hands ``padding + padding`` = (ph, pw, ph, pw)
    to F.pad, whose order is (left, right, top, bottom) -- so W is padded by ph on the left and pw on the right, H by ph at
    the top and pw at the bottom (meta_conv.py:159, 175-176).  Identical to the intended padding whenever ph == pw, which is
    what every reference configuration uses; the golden vectors of tests/golden/meta_conv2d_general.npz pin the rest."""
    assert x.shape[0] == w.shape[0]
    kh, kw = (kernel_size, kernel_size) if isinstance(kernel_size, int) else kernel_size
    pad = (padding, padding) if isinstance(padding, int) else tuple(padding)
    cin = x.shape[1]
    outs = []
    for b in range(x.shape[0]):
        xb = x[b:b + 1]
        if padding_mode != 'zeros' and any(pad):
            xb = pad2d(xb, (pad[0], pad[1], pad[0], pad[1]), padding_mode)       # (top, bottom, left, right): see above
            p = 0
        else:
            p = pad
        wb = w[b].reshape(out_channels, cin // groups, kh, kw)
        outs.append(F.conv2d(xb, wb, None, stride=stride, padding=p, dilation=dilation, groups=groups))
    return torch.cat(outs, dim=0)


# --------------------------------------------------------------------------------------
# Op A: k=1 patch conv
# --------------------------------------------------------------------------------------
def patch_conv_k1(x, wt, out_channels, groups=1):
    """y[b,o,y,x] = sum_c Wt[b, o*Cin_g + c, i, j] * in[b, grp(o)*Cin_g + c, y, x],
    (i, j) = (y // ph, x // pw).  x (B,Cin,H,W), wt (B,hp,fh,fw)."""
    b, cin, h, w = x.shape
    fh, fw = wt.shape[-2:]
    assert h % fh == 0 and w % fw == 0, 'feature map must tile exactly into the weight grid'
    ph, pw = h // fh, w // fw
    cin_g, cout_g = cin // groups, out_channels // groups
    assert wt.shape[1] == out_channels * cin_g
    xv = x.reshape(b, groups, cin_g, fh, ph, fw, pw)
    wv = wt.reshape(b, groups, cout_g, cin_g, fh, fw)
    y = torch.einsum('bgocij,bgciujv->bgoiujv', wv, xv)
    return y.reshape(b, out_channels, h, w)


# --------------------------------------------------------------------------------------
# Op B: kxk patch conv with image-level padding (stride 1, dilation 1 as in every config)
# --------------------------------------------------------------------------------------
def patch_conv_kxk(x, wt, out_channels, kernel_size, padding, padding_mode='reflect', groups=1):
    """n = ((o*Cin_g + c)*k + ky)*k + kx;
    y[b,o,y,x] = sum_{c,ky,kx} Wt[b,n,i,j] * xp[b, grp(o)*Cin_g + c, y+ky, x+kx] with xp the padded
    WHOLE image: halo pixels come from the neighbouring patch's input, filtered with the current
    patch's weights."""
    b, cin, h, w = x.shape
    fh, fw = wt.shape[-2:]
    k = kernel_size
    assert 2 * padding == k - 1, 'only "same" padding is exercised by the reference configs'
    assert h % fh == 0 and w % fw == 0
    ph, pw = h // fh, w // fw
    cin_g, cout_g = cin // groups, out_channels // groups
    assert wt.shape[1] == out_channels * cin_g * k * k
    xp = pad2d(x, (padding, padding), padding_mode)
    wv = wt.reshape(b, groups, cout_g, cin_g, k, k, fh, fw)
    y = x.new_zeros(b, groups, cout_g, fh, ph, fw, pw)
    for ky in range(k):
        for kx in range(k):
            xs = xp[:, :, ky:ky + h, kx:kx + w].reshape(b, groups, cin_g, fh, ph, fw, pw)
            y = y + torch.einsum('bgocij,bgciujv->bgoiujv', wv[:, :, :, :, ky, kx], xs)
    return y.reshape(b, out_channels, h, w)


def meta_patch_conv2d(x, wt, out_channels, kernel_size, padding=0, padding_mode='reflect', groups=1):
    """MetaPatchConv2d.forward / HyperPatchConv2d core (a3/a5): Op A when k=1, else Op B."""
    if kernel_size == 1:
        assert padding == 0
        return patch_conv_k1(x, wt, out_channels, groups)
    return patch_conv_kxk(x, wt, out_channels, kernel_size, padding, padding_mode, groups)


# --------------------------------------------------------------------------------------
# Op C: v1_0 fused inverted residual (every patch independent on its own halo tile)
# --------------------------------------------------------------------------------------
def gather_halo_tiles(x, fh, fw, pad, mode='reflect'):
    """(B,C,H,W) -> (B,C,fh,fw,ph+2p,pw+2p): tile (i,j)[u,v] = padded[i*ph+u, j*pw+v]."""
    b, c, h, w = x.shape
    ph, pw = h // fh, w // fw
    xp = pad2d(x, (pad, pad), mode)
    ys = (torch.arange(fh).view(fh, 1) * ph + torch.arange(ph + 2 * pad).view(1, -1))   # fh x th
    xs = (torch.arange(fw).view(fw, 1) * pw + torch.arange(pw + 2 * pad).view(1, -1))   # fw x tw
    t = xp[:, :, ys, :]                  # B C fh th Wp
    t = t[:, :, :, :, xs]                # B C fh th fw tw
    return t.permute(0, 1, 2, 4, 3, 5)   # B C fh fw th tw


def patch_inverted_residual_v1(x, wt, hidden, out_channels, bn1, bn2, bn3, training=False):
    """Op C (hyperseg_v1_0.py:328-376).  Flat ranges [0,Cin*hid) | [+hid*9) | [+hid*Cout).

    h1 = relu6(bn1(W1 . tile))   on the (ph+2)x(pw+2) halo tile, W1[h,c] = Wt[h*Cin + c]
    h2 = relu6(bn2(dw3x3_valid(h1))), K[h,ky,kx] = Wt[Cin*hid + h*9 + ky*3 + kx]
    out = bn3(W3 . h2),            W3[o,h] = Wt[Cin*hid + 9*hid + o*hid + h]
    (+ x if Cin == Cout).  In training mode returns (out, new_stats) with BN1 statistics over
    the duplicated halo pixels (Appendix D-4)."""
    b, cin, h, w = x.shape
    fh, fw = wt.shape[-2:]
    assert h % fh == 0 and w % fw == 0
    ph, pw = h // fh, w // fw
    r1 = cin * hidden
    r2 = r1 + hidden * 9
    r3 = r2 + hidden * out_channels
    assert wt.shape[1] == r3
    w1 = wt[:, :r1].reshape(b, hidden, cin, fh, fw)
    kd = wt[:, r1:r2].reshape(b, hidden, 3, 3, fh, fw)
    w3 = wt[:, r2:r3].reshape(b, out_channels, hidden, fh, fw)

    tiles = gather_halo_tiles(x, fh, fw, 1, 'reflect')                 # B C fh fw th tw
    h1 = torch.einsum('bhcij,bcijuv->bhijuv', w1, tiles)
    stats = {}
    if training:
        h1, stats['bn1.running_mean'], stats['bn1.running_var'] = bn_train(h1, bn1)
    else:
        h1 = bn_eval(h1, bn1)
    h1 = act(h1, ACT_RELU6)
    h2 = x.new_zeros(b, hidden, fh, fw, ph, pw)
    for ky in range(3):
        for kx in range(3):
            h2 = h2 + kd[:, :, ky, kx, :, :, None, None] * h1[..., ky:ky + ph, kx:kx + pw]
    if training:
        h2, stats['bn2.running_mean'], stats['bn2.running_var'] = bn_train(h2, bn2)
    else:
        h2 = bn_eval(h2, bn2)
    h2 = act(h2, ACT_RELU6)
    out = torch.einsum('bohij,bhijuv->boijuv', w3, h2)
    if training:
        out, stats['bn3.running_mean'], stats['bn3.running_var'] = bn_train(out, bn3)
    else:
        out = bn_eval(out, bn3)
    out = out.permute(0, 1, 2, 4, 3, 5).reshape(b, out_channels, h, w)
    if cin == out_channels:
        out = x + out
    return (out, stats) if training else out


# --------------------------------------------------------------------------------------
# Op D: v0_1 inverted residual = three image-level patch convs
# --------------------------------------------------------------------------------------
def patch_inverted_residual_v0(x, wt, hidden, out_channels, bn1, bn2, bn3):
    """relu6(bn(OpA_pw1(x))) -> relu6(bn(OpB_dw3x3_reflect(.))) -> bn(OpA_pw3(.)), weights from
    consecutive channel ranges (hyperseg_v0_1.py:205-237).  Not equal to Op C at patch borders."""
    cin = x.shape[1]
    r1 = cin * hidden
    r2 = r1 + hidden * 9
    r3 = r2 + hidden * out_channels
    assert wt.shape[1] == r3
    # expand_ratio != 1 in every reference config (the pw1 block exists iff expand_ratio != 1)
    y = act(bn_eval(patch_conv_k1(x, wt[:, :r1], hidden), bn1), ACT_RELU6)
    y = act(bn_eval(patch_conv_kxk(y, wt[:, r1:r2], hidden, 3, 1, 'reflect', groups=hidden), bn2), ACT_RELU6)
    y = bn_eval(patch_conv_k1(y, wt[:, r2:r3], out_channels), bn3)
    if cin == out_channels:
        y = x + y
    return y


# --------------------------------------------------------------------------------------
# Op E: stage glue
# --------------------------------------------------------------------------------------
def stage_input(skip, prev):
    """cat(coords(2), skip, up(prev)) in that channel order (hyperseg_v1_0.py:231-240)."""
    b, _, h, w = skip.shape
    parts = [image_coords(h, w, skip.dtype).unsqueeze(0).expand(b, -1, -1, -1), skip]
    if prev is not None:
        if prev.shape[-2:] != skip.shape[-2:]:
            prev = upsample_bilinear(prev, (h, w))
        parts.append(prev)
    return torch.cat(parts, dim=1)


# --------------------------------------------------------------------------------------
# constructor arithmetic restated (what init does that the forward depends on)
# --------------------------------------------------------------------------------------
def divide_feature(in_feature, out_features, min_unit=8):
    """Split ``in_feature`` channels between consumers proportionally to ``out_features`` in
    multiples of ``min_unit``; equal consumers get equal shares; the last (smallest-total)
    group takes the remainder.  Restates hyperseg_v1_0.py:763-810 (float floor-division kept)."""
    assert in_feature % min_unit == 0
    units = in_feature // min_unit
    order = np.argsort(out_features)
    svals = np.array(out_features)[order]
    groups = [(val, order[list(idx)]) for val, idx in groupby(range(len(order)), lambda i: svals[i])]
    groups.sort(key=lambda g: g[0] * len(g[1]), reverse=True)
    ratio = float(units) / sum(out_features)
    share = [len(members) for _, members in groups]
    left = units - sum(share)
    for gi, (val, members) in enumerate(groups):
        if gi == len(groups) - 1:
            share[-1] += left
            break
        n = len(members)
        want = max(val * n * ratio, n)
        want = want // n * n - n
        want = min(want, left)
        share[gi] += want
        left -= want
        if left == 0:
            break
    out = np.zeros(len(out_features), dtype=int)
    for gi, (_, members) in enumerate(groups):
        for m in members:
            out[m] = share[gi] // len(members) * min_unit
    return out


def divide_feature_legacy(in_feature, out_features, min_unit=8):
    """The older split used by v0_1's Conv2dMulti (hyperseg_v0_1.py:366-406)."""
    assert in_feature % min_unit == 0
    units = in_feature // min_unit
    order = np.argsort(out_features)
    svals = np.array(out_features)[order]
    groups = [(val, order[list(idx)]) for val, idx in groupby(range(len(order)), lambda i: svals[i])]
    groups.sort(key=lambda g: g[0] * len(g[1]), reverse=True)
    ratio = float(units) / sum(out_features)
    left = units
    share = []
    for gi, (val, members) in enumerate(groups):
        if gi == len(groups) - 1:
            share.append(left)
            break
        n = len(members)
        want = max(val * n * ratio, 1)
        want = want // n * n
        share.append(want)
        left -= want
    out = np.zeros(len(out_features), dtype=int)
    for gi, (_, members) in enumerate(groups):
        for m in members:
            out[m] = share[gi] // len(members) * min_unit
    return out


def decoder_plan(variant, feat_channels, signal_channels, num_classes, kernel_sizes, level_channels=None,
                 expand_ratio=1, weight_groups=1, unify_level=None):
    """Static description of a decoder: per level (k, cin, cout, hidden, hp) plus, for v1_0 and
    unify, the signal2weights layers (signal_index, signal_channels, groups, rows).

    feat_channels is [in_nc] + backbone.feat_channels[:-1] (fine -> coarse), as HyperGen passes it
    (hyperseg_v1_0.py:41-45).  Quirks kept: v1_0's signal_index is 0 for every level
    (Appendix D-1); unify's offsets are cumulative; v0_1 has no level_channels (Appendix D-13)."""
    fc = list(feat_channels)[::-1]
    n_levels = len(fc) if variant == 'v0_1' else len(level_channels)
    if isinstance(kernel_sizes, int):
        kernel_sizes = [kernel_sizes] * n_levels
    levels = []
    prev = 0
    for l in range(n_levels):
        cur = fc[l]
        cout = cur if (variant == 'v0_1' or level_channels is None) else level_channels[l]
        prev += cur
        if l == n_levels - 1:
            cout = num_classes
        cin = prev + 2
        k = kernel_sizes[l]
        if k > 1:
            hid = int(round(cin * expand_ratio))
            hp = cin * hid + hid * k * k + hid * cout
        else:
            hid = 0
            hp = cout * cin
        levels.append(dict(k=k, cin=cin, cout=cout, hidden=hid, hp=hp, skip=cur))
        prev = cout
    plan = dict(variant=variant, levels=levels)
    if variant == 'v0_1':
        return plan
    if isinstance(weight_groups, int):
        wg = [weight_groups] * n_levels
    else:
        wg = list(weight_groups)
    if variant == 'v1_0':
        hps = [lv['hp'] for lv in levels]
        split = divide_feature(signal_channels, hps, min_unit=max(wg))
        plan['s2w'] = [dict(signal_index=0, signal_channels=int(split[l]), groups=wg[l],
                            rows=next_multiply(hps[l], wg[l]), hp=hps[l]) for l in range(n_levels)]
    elif variant == 'unify':
        targets = [levels[l]['hp'] for l in range(unify_level - 1)]
        targets.append(sum(levels[l]['hp'] for l in range(unify_level - 1, n_levels)))
        split = divide_feature(signal_channels, targets, min_unit=max(wg))
        s2w, off = [], 0
        for i, t in enumerate(targets):
            s2w.append(dict(signal_index=off, signal_channels=int(split[i]), groups=wg[i],
                            rows=next_multiply(t, wg[i]), hp=t))
            off += int(split[i])
        plan['s2w'] = s2w
        plan['unify_level'] = unify_level
        ranges = [0]
        for l in range(unify_level - 1, n_levels):
            ranges.append(ranges[-1] + levels[l]['hp'])
        plan['unify_ranges'] = ranges
    else:
        raise ValueError(variant)
    return plan


def _bn(params, prefix):
    return {k: params[f'{prefix}.{k}'] for k in ('weight', 'bias', 'running_mean', 'running_var')}


# --------------------------------------------------------------------------------------
# a8: the three decoders.  ``params`` uses the reference's state-dict key names (Appendix C),
# without the leading "decoder." prefix.
# --------------------------------------------------------------------------------------
def decoder_v1_0(plan, params, x, s, return_levels=False, training=False):
    """MultiScaleDecoder.forward of hyperseg_v1_0.py:221-253.  x: list fine->coarse incl. image.
    ``training=True``: BatchNorm with batch statistics (BN1 of the inverted residual over the duplicated halo pixels,
    Appendix D-4); returns (logits, new_running_stats) -- everything is plain torch, so autograd of this function is
    the oracle for the backward kernels (checked against the reference's own gradients in the train_* fixtures)."""
    p, outs, stats = None, [], {}
    for l, lv in enumerate(plan['levels']):
        inp = stage_input(x[-l - 1], p)
        sw = plan['s2w'][l]
        if lv['k'] == 1:
            wt = signal2weights(s, params[f'level_{l}.0.0.signal2weights.weight'], sw['signal_index'],
                                sw['signal_channels'], sw['groups'], lv['hp'])
            y = patch_conv_k1(inp, wt, lv['cout'])
            if training:
                y, m, v = bn_train(y, _bn(params, f'level_{l}.0.1'))
                stats[f'level_{l}.0.1.running_mean'], stats[f'level_{l}.0.1.running_var'] = m, v
            else:
                y = bn_eval(y, _bn(params, f'level_{l}.0.1'))
            p = act(y, ACT_RELU)
        else:
            wt = signal2weights(s, params[f'level_{l}.0.signal2weights.weight'], sw['signal_index'],
                                sw['signal_channels'], sw['groups'], lv['hp'])
            p = patch_inverted_residual_v1(inp, wt, lv['hidden'], lv['cout'], _bn(params, f'level_{l}.0.bn1'),
                                           _bn(params, f'level_{l}.0.bn2'), _bn(params, f'level_{l}.0.bn3'),
                                           training=training)
            if training:
                p, st = p
                stats.update({f'level_{l}.0.{k}': v for k, v in st.items()})
        outs.append(p)
    if p.shape[-2:] != x[0].shape[-2:]:
        p = upsample_bilinear(p, x[0].shape[-2:])
    if training:
        return p, stats
    return (p, outs) if return_levels else p


def decoder_unify(plan, params, x, s, return_levels=False):
    """MultiScaleDecoder.forward of hyperseg_v1_0_unify.py:222-259."""
    p, outs = None, []
    ul = plan['unify_level']
    w_shared = None
    for l, lv in enumerate(plan['levels']):
        inp = stage_input(x[-l - 1], p)
        wi = min(l, ul - 1)
        sw = plan['s2w'][wi]
        if l < ul - 1:
            wt = signal2weights(s, params[f'weight_blocks.{wi}.signal2weights.weight'], sw['signal_index'],
                                sw['signal_channels'], sw['groups'], sw['hp'])
        else:
            if l == ul - 1:
                w_shared = signal2weights(s, params[f'weight_blocks.{wi}.signal2weights.weight'],
                                          sw['signal_index'], sw['signal_channels'], sw['groups'], sw['hp'])
            i = l - ul + 1
            wt = w_shared[:, plan['unify_ranges'][i]:plan['unify_ranges'][i + 1]]
        if lv['k'] == 1:
            y = patch_conv_k1(inp, wt[:, :lv['hp']], lv['cout'])
            p = act(bn_eval(y, _bn(params, f'level_blocks.{l}.0.1')), ACT_RELU)
        else:
            p = patch_inverted_residual_v1(inp, wt[:, :lv['hp']], lv['hidden'], lv['cout'],
                                           _bn(params, f'level_blocks.{l}.0.bn1'),
                                           _bn(params, f'level_blocks.{l}.0.bn2'),
                                           _bn(params, f'level_blocks.{l}.0.bn3'))
        outs.append(p)
    if p.shape[-2:] != x[0].shape[-2:]:
        p = upsample_bilinear(p, x[0].shape[-2:])
    return (p, outs) if return_levels else p


def decoder_v0_1(plan, params, x, w, return_levels=False):
    """MultiScaleDecoder.forward of hyperseg_v0_1.py:173-202.  w: list of per-level (B,hp,fh,fw)."""
    p, outs = None, []
    for l in range(len(x)):
        lv = plan['levels'][l]
        inp = stage_input(x[-l - 1], p)
        if lv['k'] == 1:
            y = patch_conv_k1(inp, w[l][:, :lv['hp']], lv['cout'])
            p = act(bn_eval(y, _bn(params, f'level_{l}.0.1')), ACT_RELU)
        else:
            p = patch_inverted_residual_v0(inp, w[l][:, :lv['hp']], lv['hidden'], lv['cout'],
                                           _bn(params, f'level_{l}.0.conv.0.1'),
                                           _bn(params, f'level_{l}.0.conv.1.1'),
                                           _bn(params, f'level_{l}.0.conv.2.1'))
        outs.append(p)
    return (p, outs) if return_levels else p


# --------------------------------------------------------------------------------------
# synthetic decoder workloads of SURVEY.md section 8(d) (seeded; used by tests and bench)
# --------------------------------------------------------------------------------------
CONFIGS = {
    # name: variant, image HxW, feat_channels (image + backbone taps, fine->coarse), signal ch, kwargs
    'M': dict(variant='v1_0', size=(512, 1024), feat=[3, 16, 6, 10, 28, 80], signal=1280, num_classes=19,
              kernel_sizes=[1, 1, 1, 3, 3], level_channels=[64, 32, 16, 16, 16], expand_ratio=2,
              weight_groups=[32, 16, 8, 16, 4]),
    'S': dict(variant='unify', size=(768, 1536), feat=[3, 16, 4, 8, 28, 128], signal=1280, num_classes=19,
              kernel_sizes=[1, 1, 1, 3, 3], level_channels=[32, 16, 8, 8, 8], expand_ratio=2,
              weight_groups=[32, 16, 8, 16, 4], unify_level=4),
    'Sc': dict(variant='v1_0', size=(576, 768), feat=[3, 4, 6, 10, 28, 80], signal=1280, num_classes=12,
               kernel_sizes=[1, 1, 1, 3, 3], level_channels=[64, 32, 16, 16, 16], expand_ratio=2,
               weight_groups=[64, 32, 32, 16, 8]),
    'L': dict(variant='v0_1', size=(512, 512), feat=[3, 6, 8, 12, 34, 96], signal=1536, num_classes=21,
              kernel_sizes=[1, 1, 3, 3, 3, 3], expand_ratio=2),
    # CamVid HyperSeg-L (configs/train/camvid_efficientnet_b1_hyperseg-l.py:35-38; not a BASELINE config): the only shipped v1_0
    # decoder with SIX levels -- three k = 1, then three inverted residuals, the last on 32 x 32-pixel patches of the full-resolution
    # image (the raw image is its skip) -- evaluated at 1024 x 768 (val_img_transforms, :21); backbone taps as CamVid-S
    'Lc': dict(variant='v1_0', size=(768, 1024), feat=[3, 4, 6, 10, 28, 80], signal=1280, num_classes=12,
               kernel_sizes=[1, 1, 1, 3, 3, 3], level_channels=[64, 32, 16, 16, 16, 16], expand_ratio=2,
               weight_groups=[64, 32, 32, 16, 8, 8]),
}


def config_plan(name):
    c = CONFIGS[name]
    return decoder_plan(c['variant'], c['feat'], c['signal'], c['num_classes'], c['kernel_sizes'],
                        c.get('level_channels'), c['expand_ratio'], c.get('weight_groups', 1),
                        c.get('unify_level'))


def level_row_fans(lv):
    """Fan-in of every flat weight row of a level (used to keep synthetic activations O(1))."""
    if lv['k'] == 1:
        return [lv['cin']] * lv['hp']
    return [lv['cin']] * (lv['cin'] * lv['hidden']) + [9] * (lv['hidden'] * 9) + \
           [lv['hidden']] * (lv['hidden'] * lv['cout'])


def synth_bn(gen, n):
    return {
        'weight': torch.rand(n, generator=gen) + 0.5,
        'bias': torch.randn(n, generator=gen) * 0.1,
        'running_mean': torch.randn(n, generator=gen) * 0.1,
        'running_var': torch.rand(n, generator=gen) * 1.5 + 0.5,
    }


def synth_decoder_params(plan, seed=0):
    """Seeded decoder parameters: signal2weights ~ N(0, 1/fan_in), BN stats/affine randomised
    (SURVEY.md section 8d), keyed like the reference state dict."""
    gen = torch.Generator().manual_seed(seed)
    params = {}
    variant = plan['variant']

    def put_bn(prefix, n):
        for k, v in synth_bn(gen, n).items():
            params[f'{prefix}.{k}'] = v

    for l, lv in enumerate(plan['levels']):
        lvl = f'level_blocks.{l}' if variant == 'unify' else f'level_{l}'
        if lv['k'] == 1:
            put_bn(f'{lvl}.0.1', lv['cout'])
        elif variant == 'v0_1':
            put_bn(f'{lvl}.0.conv.0.1', lv['hidden'])
            put_bn(f'{lvl}.0.conv.1.1', lv['hidden'])
            put_bn(f'{lvl}.0.conv.2.1', lv['cout'])
        else:
            put_bn(f'{lvl}.0.bn1', lv['hidden'])
            put_bn(f'{lvl}.0.bn2', lv['hidden'])
            put_bn(f'{lvl}.0.bn3', lv['cout'])
    for i, sw in enumerate(plan.get('s2w', [])):
        fan_in = sw['signal_channels'] // sw['groups']
        if variant == 'unify' and i == plan['unify_level'] - 1:
            fans = sum((level_row_fans(lv) for lv in plan['levels'][i:]), [])
        else:
            fans = level_row_fans(plan['levels'][i])
        fans = torch.tensor(fans + [fans[-1]] * (sw['rows'] - len(fans)), dtype=torch.float32)
        # E[s^2] = 0.5 for s = relu(N(0,1)); choose std so that E[Wt^2] = 1 / fan_row
        std = torch.sqrt(2.0 / (fan_in * fans)).view(-1, 1, 1, 1)
        wgt = torch.randn(sw['rows'], fan_in, 1, 1, generator=gen) * std
        if variant == 'unify':
            params[f'weight_blocks.{i}.signal2weights.weight'] = wgt
        elif plan['levels'][i]['k'] == 1:
            params[f'level_{i}.0.0.signal2weights.weight'] = wgt
        else:
            params[f'level_{i}.0.signal2weights.weight'] = wgt
    return params


def synth_decoder_inputs(name_or_cfg, batch=1, seed=0, size=None):
    """Skip features ~ N(0,1) at the exact strides (1,2,4,8,16,32), signal = relu(N(0,1)) at /32
    (v1_0 / unify), or per-level weight tensors ~ N(0, 1/fan_in) (v0_1)."""
    c = CONFIGS[name_or_cfg] if isinstance(name_or_cfg, str) else name_or_cfg
    h, w = size if size is not None else c['size']
    gen = torch.Generator().manual_seed(seed + 1000)
    x = [torch.randn(batch, ch, h >> i, w >> i, generator=gen) for i, ch in enumerate(c['feat'])]
    fh, fw = h // 32, w // 32
    if c['variant'] == 'v0_1':
        plan = decoder_plan(c['variant'], c['feat'], c['signal'], c['num_classes'], c['kernel_sizes'],
                            None, c['expand_ratio'])
        ws = []
        for lv in plan['levels']:
            std = torch.rsqrt(torch.tensor(level_row_fans(lv), dtype=torch.float32)).view(1, -1, 1, 1)
            ws.append(torch.randn(batch, lv['hp'], fh, fw, generator=gen) * std)
        return x, ws
    s = torch.randn(batch, c['signal'], fh, fw, generator=gen).clamp(min=0)
    return x, s


def run_config(name, batch=1, seed=0, size=None, dtype=torch.float32, return_levels=False):
    """Build plan + synthetic params + inputs of a named BASELINE config and run the oracle."""
    plan = config_plan(name)
    params = {k: v.to(dtype) for k, v in synth_decoder_params(plan, seed).items()}
    x, sw = synth_decoder_inputs(name, batch, seed, size)
    x = [t.to(dtype) for t in x]
    fn = {'v1_0': decoder_v1_0, 'unify': decoder_unify, 'v0_1': decoder_v0_1}[plan['variant']]
    sw = [t.to(dtype) for t in sw] if isinstance(sw, list) else sw.to(dtype)
    return fn(plan, params, x, sw, return_levels=return_levels)
