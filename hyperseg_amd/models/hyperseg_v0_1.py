"""HyperSeg v0.1 on the MI355X decoder path -- drop-in for hyperseg/models/hyperseg_v0_1.py
(HyperSeg-L PASCAL VOC: configs/train/vocsbd_efficientnet_b3_hyperseg-l.py:10, 32-34).

The oldest variant: the context head (:class:`WeightMapper` + :class:`Conv2dMulti`, stock PyTorch) emits a LIST of
per-level weight tensors (B, hp_l, H/32, W/32); the decoder is built purely from the ``layers/`` modules, has one level
per pyramid entry INCLUDING the input image (so no final upsample), no ``level_channels`` (Appendix D-13), and its
inverted residual is three IMAGE-level patch convolutions (hyperseg_v0_1.py:205-237: the depthwise conv's halo comes from
the neighbouring patch's hidden activations -- Op D, not Op C).  Each conv + BatchNorm + ReLU6 block is one
``hs_patch_conv_fwd`` launch (via MetaSequential's fusion) with the level's stage input generated in the first one.
"""
from functools import partial
from itertools import groupby

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as HF
from ._common import EpochOnModeSwitch, HyperGenBase, coordinate_grid, per_level
from .layers.meta_patch import MetaPatchConv2d, make_meta_patch_conv2d_block
from .layers.meta_sequential import MetaSequential


def next_multiply(x, base):
    return type(x)(np.ceil(x / base) * base)


def get_image_coordinates(b, h, w, device):
    """(b, 2, h, w) coordinate channels (hyperseg_v0_1.py:240-246); API parity -- the kernels generate them in place."""
    return coordinate_grid(h, w, device).repeat(b, 1, 1, 1)


class HyperPatchInvertedResidual(EpochOnModeSwitch, nn.Module):
    """pw1 (+BN+ReLU6) -> depthwise kxk reflect (+BN+ReLU6) -> pw-linear (+BN), each a patch-wise dynamic conv on the
    whole image, weights taken from consecutive channel ranges by the MetaSequential (hyperseg_v0_1.py:205-237)."""

    def __init__(self, in_nc, out_nc, kernel_size=3, stride=1, expand_ratio=1, norm_layer=nn.BatchNorm2d,
                 act_layer=nn.ReLU6(inplace=True), padding_mode='reflect'):
        super(HyperPatchInvertedResidual, self).__init__()
        self.stride = stride
        assert stride in [1, 2]
        hidden_dim = int(round(in_nc * expand_ratio))
        self.use_res_connect = self.stride == 1 and in_nc == out_nc
        layers = []
        if expand_ratio != 1:
            layers.append(make_meta_patch_conv2d_block(in_nc, hidden_dim, 1, norm_layer=norm_layer, act_layer=act_layer))
        layers.extend([
            make_meta_patch_conv2d_block(hidden_dim, hidden_dim, kernel_size, stride=stride, groups=hidden_dim,
                                         norm_layer=norm_layer, act_layer=act_layer, padding_mode=padding_mode),
            make_meta_patch_conv2d_block(hidden_dim, out_nc, 1, stride=stride, norm_layer=norm_layer, act_layer=None)
        ])
        self.conv = MetaSequential(*layers)

    @property
    def hyper_params(self):
        return self.conv.hyper_params

    def _fused_parts(self):
        """(pw1, bn1, dw, bn2, pw3, bn3) if the block has exactly the structure hs_patch_ir_v0_fwd implements
        (expand_ratio != 1, 3x3 depthwise with reflect padding, BatchNorm2d, ReLU6, no residual), else None."""
        blocks = list(self.conv)
        if self.use_res_connect or len(blocks) != 3 or any(len(b) < 2 for b in blocks):
            return None
        (c1, n1, *r1), (c2, n2, *r2), (c3, n3, *r3) = [list(b) for b in blocks]
        ok = all(isinstance(c, MetaPatchConv2d) for c in (c1, c2, c3)) and \
            all(isinstance(n, nn.BatchNorm2d) for n in (n1, n2, n3)) and \
            len(r1) == 1 and isinstance(r1[0], nn.ReLU6) and len(r2) == 1 and isinstance(r2[0], nn.ReLU6) and not r3 and \
            c1.kernel_size == (1, 1) and c3.kernel_size == (1, 1) and c1.groups == 1 and c3.groups == 1 and \
            c2.kernel_size == (3, 3) and c2.groups == c2.in_channels == c2.out_channels and \
            c2.padding == (1, 1) and c2.padding_mode == 'reflect' and \
            all(c.hyper_module.stride == (1, 1) and c.hyper_module.dilation == (1, 1) for c in (c1, c2, c3))
        return (c1, n1, c2, n2, c3, n3) if ok else None

    def _forward_fused(self, x, w):
        parts = self._fused_parts()
        if parts is None or not isinstance(x, HF.StageInput):
            return None
        c1, n1, c2, n2, c3, n3 = parts
        from .. import autograd as HA
        if any(n.training for n in (n1, n2, n3)) or HA.needs_grad(w, x.skip, x.prev, n1.weight, n2.weight, n3.weight):
            return None
        hp = int(self.hyper_params)
        if w.dim() != 4 or w.shape[1] < hp:
            raise ValueError(f'weight must be (B, >={hp}, fh, fw), got {tuple(w.shape)}')
        if getattr(self, '_folded', None) is None:
            self._folded = [HF.FoldedBN(), HF.FoldedBN(), HF.FoldedBN()]
        bns = [f.get(n) for f, n in zip(self._folded, (n1, n2, n3))]
        if isinstance(w, HF.BankRef):
            bank = w.bank
        else:
            # the level's weights arrive channel-major (B, hp, fh, fw): one re-layout for the whole block
            bank = HF.bank_pack(w, 0, hp)
        return HF.patch_ir_v0(x, tuple(w.shape[-2:]), bank, c1.out_channels, c3.out_channels, *bns,
                              math=getattr(self, 'ir_math', None))

    def forward(self, x, w):
        y = self._forward_fused(x, w)          # one launch (Op D) for the decoder's shapes
        if y is not None:
            return y
        if self.use_res_connect:
            xin = x.materialize() if isinstance(x, HF.StageInput) else x
            return xin + self.conv(xin, w)
        return self.conv(x, w)


class MultiScaleDecoder(EpochOnModeSwitch, nn.Module):
    """hyperseg_v0_1.py:91-202.  ``forward(x, w)``: x fine -> coarse incl. the image, w = list of per-level weights."""

    def __init__(self, feat_channels, in_nc=3, num_classes=3, kernel_sizes=3, level_layers=1, norm_layer=nn.BatchNorm2d,
                 act_layer=nn.ReLU6(inplace=True), out_kernel_size=1, expand_ratio=1, with_out_fc=False, dropout=None):
        super(MultiScaleDecoder, self).__init__()
        n = len(feat_channels)
        kernel_sizes = per_level(kernel_sizes, n, 'kernel_sizes')
        level_layers = per_level(level_layers, n, 'level_layers')
        self.level_layers, self.levels = level_layers, n
        self.layer_params = []
        coarse_to_fine = feat_channels[::-1]

        carried = 0                                   # channels handed up from the coarser level
        for lvl, (skip_nc, k, depth) in enumerate(zip(coarse_to_fine, kernel_sizes, level_layers)):
            width, carried = skip_nc, carried + skip_nc
            blocks = []
            for j in range(depth):
                if not with_out_fc and lvl == n - 1 and j == depth - 1:
                    width = num_classes               # the very last layer emits the logits
                if k > 1:
                    blocks.append(HyperPatchInvertedResidual(carried + 2, width, k, expand_ratio=expand_ratio,
                                                             norm_layer=norm_layer, act_layer=act_layer))
                else:
                    blocks.append(make_meta_patch_conv2d_block(carried + 2, width, k))
                carried = width
            self.add_module(f'level_{lvl}', MetaSequential(*blocks))

        self.out_fc = None
        if with_out_fc:
            tail = [] if dropout is None else [nn.Dropout2d(dropout, True)]
            tail.append(MetaPatchConv2d(carried, num_classes, out_kernel_size, padding=out_kernel_size // 2))
            self.out_fc = MetaSequential(*tail)

        # bookkeeping the context head sizes itself from
        self.param_groups = [getattr(self, f'level_{lvl}').hyper_params for lvl in range(n)]
        if self.out_fc is not None:
            self.param_groups.append(self.out_fc.hyper_params)
        self._ranges = [0] + list(np.cumsum(self.param_groups))
        self.hyper_params = int(self._ranges[-1])
        self._ranges.append(self.hyper_params)

    def forward(self, x, w, masks=False):
        assert isinstance(w, (list, tuple))
        assert len(x) <= self.levels
        p = None
        for level in range(len(x)):
            stage = HF.StageInput(x[-level - 1], p, coords=True)      # cat(coords, skip, bilinear(p)), never built
            p = getattr(self, f'level_{level}')(stage, w[level])
        if self.out_fc is not None:
            p = self.out_fc(p, w[-1])
        if masks and not (self.training or p.requires_grad):
            return HF.upsample_argmax(p.contiguous(), p.shape[2:])       # identity resize: argmax over classes only
        return p


def divide_feature_legacy(in_feature, out_features, min_unit=8):
    """The older channel split used by :class:`Conv2dMulti` (hyperseg_v0_1.py:366-406; "contains bugs" but the
    released HyperSeg-L checkpoint depends on its output, so it is reproduced exactly)."""
    assert in_feature % min_unit == 0, f'in_feature ({in_feature}) must be divisible by min_unit ({min_unit})'
    units = in_feature // min_unit
    order = np.argsort(out_features)
    sorted_vals = np.array(out_features)[order]
    groups = [(val, order[list(idx)]) for val, idx in groupby(range(len(order)), lambda i: sorted_vals[i])]
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
        want = want // n * n                      # float floor-division, as the reference
        share.append(want)
        left -= want
    out = np.zeros(len(out_features), dtype=int)
    for gi, (_, members) in enumerate(groups):
        for m in members:
            out[m] = share[gi] // len(members) * min_unit
    return out


class Conv2dMulti(nn.Module):
    """One Conv2d per consumer, each reading its own slice of the input channels (hyperseg_v0_1.py:336-362)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, padding_mode='zeros', min_unit=8):
        super(Conv2dMulti, self).__init__()
        self.in_channels, self.out_channels, self.bias = in_channels, out_channels, bias
        widths = [int(v) for v in divide_feature_legacy(in_channels, out_channels, min_unit)]
        self._ranges = [0] + [int(v) for v in np.cumsum(widths)]
        for i, (cin, cout) in enumerate(zip(widths, out_channels)):
            self.add_module(f'conv_{i}', nn.Conv2d(cin, int(cout), kernel_size, stride, padding, dilation, groups, bias,
                                                   padding_mode))

    def forward(self, x):
        lo, hi = self._ranges[:-1], self._ranges[1:]
        return [getattr(self, f'conv_{i}')(x[:, a:b]) for i, (a, b) in enumerate(zip(lo, hi))]

    def extra_repr(self):
        return f'in_channels={self.in_channels}, out_channels={self.out_channels}, bias={self.bias}'


class WeightMapper(nn.Module):
    """v0_1 context head (hyperseg_v0_1.py:249-329): stride-2 2x2 convs down, optional global average at the bottom,
    nearest 2x up + concat + 1x1 "flat" merges, then :class:`Conv2dMulti` emits one weight tensor per level."""

    def __init__(self, in_channels, out_channels, levels=2, bias=False, min_unit=8, down_groups=1, flat_groups=1,
                 weight_groups=1, avg_pool=False):
        super(WeightMapper, self).__init__()
        if levels <= 0:
            raise AssertionError('levels must be greater than zero')
        self.in_channels, self.out_channels, self.levels, self.bias = in_channels, out_channels, levels, bias
        self.avg_pool, self.down_groups, self.flat_groups, self.weight_groups = avg_pool, down_groups, flat_groups, weight_groups
        nc = in_channels
        for i in range(levels - 1):
            down = nn.Sequential(nn.Conv2d(nc, nc, 2, 2, bias=bias, groups=down_groups), nn.BatchNorm2d(nc), nn.ReLU(True))
            merge = [nn.Conv2d(2 * nc, nc, 1, bias=bias, groups=flat_groups), nn.BatchNorm2d(nc)]
            if i > 0:
                merge.append(nn.ReLU(True))          # the top merge feeds out_conv un-activated
            self.add_module(f'down_{i}', down)
            self.add_module(f'up_{i}', nn.UpsamplingNearest2d(scale_factor=2))
            self.add_module(f'flat_{i}', nn.Sequential(*merge))
        padded = [next_multiply(c, weight_groups) for c in out_channels]
        self.out_conv = Conv2dMulti(nc, padded, 1, bias=bias, min_unit=max(min_unit, weight_groups), groups=weight_groups)

    def forward(self, x):
        pyramid = [x]
        for i in range(self.levels - 1):
            pyramid.append(getattr(self, f'down_{i}')(pyramid[-1]))
        if self.avg_pool and self.levels > 1 and pyramid[-1].shape[-2:] != (1, 1):
            pyramid[-1] = F.adaptive_avg_pool2d(pyramid[-1], 1).expand_as(pyramid[-1])
        while len(pyramid) > 1:
            i = len(pyramid) - 2
            coarse = getattr(self, f'up_{i}')(pyramid.pop())
            pyramid[-1] = getattr(self, f'flat_{i}')(torch.cat((pyramid[-1], coarse), dim=1))
        banks = self._banks_hip(pyramid[0])
        if banks is not None:
            return banks
        banks = self.out_conv(pyramid[0])
        if self.weight_groups > 1:                    # drop the rows added to round up to the group count
            banks = [t[:, :rows] for t, rows in zip(banks, self.out_channels)]
        return banks

    def _banks_hip(self, feat):
        """Inference: Conv2dMulti + the ``[:, :rows]`` truncation (hyperseg_v0_1.py:323-324, 336-359) as ONE
        hs_signal2weights_multi_fwd launch that writes every level's bank patch-major (f32 MFMA; the stock grouped
        1x1 convolutions fall into MIOpen's naive kernel: 36 % of HyperSeg-L's kernel time).  Returns a list of
        HF.BankRef, or None when the stock path has to run (training, biases, K = Cin/groups > 80)."""
        from .. import autograd as HA
        convs = [getattr(self.out_conv, f'conv_{i}') for i in range(len(self.out_channels))]
        if not feat.is_cuda or self.training or self.bias or HA.needs_grad(feat, *[c.weight for c in convs]):
            return None
        if any(c.kernel_size != (1, 1) or c.in_channels // c.groups > 80 for c in convs):
            return None
        if getattr(self, '_s2w_t', None) is None:
            self._s2w_t = [HF.TransposedS2W() for _ in convs]
        lo = self.out_conv._ranges
        layers = [dict(wsw_t=t.get(c), signal_index=lo[i], signal_channels=c.in_channels, groups=c.groups,
                       rows=int(self.out_channels[i])) for i, (c, t) in enumerate(zip(convs, self._s2w_t))]
        return HF.signal2weights_multi(feat.contiguous(), layers)

    def extra_repr(self):
        return f'in_channels={self.in_channels}, out_channels={self.out_channels}, bias={self.bias}'


class HyperGen(HyperGenBase):
    """hyperseg_v0_1.py:11-88; the context head returns a LIST of per-level weight tensors.  Inference modes in HyperGenBase."""

    def __init__(self, backbone, weight_mapper, in_nc=3, num_classes=3, kernel_sizes=3, level_layers=1, expand_ratio=1,
                 groups=1, inference_hflip=False, inference_gather='mean', with_out_fc=False, decoder_dropout=None):
        super(HyperGen, self).__init__()
        self.inference_hflip, self.inference_gather = inference_hflip, inference_gather
        self.backbone = backbone()
        taps = self.backbone.feat_channels
        self.decoder = MultiScaleDecoder([in_nc] + taps[:-1], 3, num_classes, kernel_sizes, level_layers,
                                         with_out_fc=with_out_fc, out_kernel_size=1, expand_ratio=expand_ratio,
                                         dropout=decoder_dropout)
        self.weight_mapper = weight_mapper(taps[-1], self.decoder.param_groups)


def hyperseg_efficientnet(model_name, pretrained=False, levels=3, down_groups=1, flat_groups=1, weight_groups=1,
                          avg_pool=True, weights_path=None, **kwargs):
    """Config-file factory (hyperseg_v0_1.py:409-424)."""
    from .backbones.efficientnet import efficientnet

    weight_mapper = partial(WeightMapper, levels=levels, down_groups=down_groups, flat_groups=flat_groups,
                            weight_groups=weight_groups, avg_pool=avg_pool)
    backbone = partial(efficientnet, model_name, pretrained=pretrained, head=None, return_features=True)
    model = HyperGen(backbone, weight_mapper, **kwargs)
    if weights_path is not None:
        checkpoint = torch.load(weights_path, map_location='cpu', weights_only=False)
        model.load_state_dict(checkpoint['state_dict'], strict=True)
    return model
