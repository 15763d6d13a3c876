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
from .layers.meta_patch import MetaPatchConv2d, make_meta_patch_conv2d_block
from .layers.meta_sequential import MetaSequential


def next_multiply(x, base):
    return type(x)(np.ceil(x / base) * base)


def get_image_coordinates(b, h, w, device):
    """(b, 2, h, w): channel 0 = x in [-1, 1], channel 1 = y (hyperseg_v0_1.py:240-246).  Kept for API parity; the
    decoder generates the same values inside the stage kernels."""
    x = torch.linspace(-1, 1, steps=w, device=device)
    y = torch.linspace(-1, 1, steps=h, device=device)
    return torch.stack([x.view(1, w).expand(h, w), y.view(h, 1).expand(h, w)], dim=0).repeat(b, 1, 1, 1)


class HyperPatchInvertedResidual(nn.Module):
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

    def forward(self, x, w):
        if self.use_res_connect:
            xin = x.materialize() if isinstance(x, HF.StageInput) else x
            return xin + self.conv(xin, w)
        return self.conv(x, w)


class MultiScaleDecoder(nn.Module):
    """hyperseg_v0_1.py:91-202.  ``forward(x, w)``: x fine -> coarse incl. the image, w = list of per-level weights."""

    def __init__(self, feat_channels, in_nc=3, num_classes=3, kernel_sizes=3, level_layers=1, norm_layer=nn.BatchNorm2d,
                 act_layer=nn.ReLU6(inplace=True), out_kernel_size=1, expand_ratio=1, with_out_fc=False, dropout=None):
        super(MultiScaleDecoder, self).__init__()
        n = len(feat_channels)
        if isinstance(kernel_sizes, int):
            kernel_sizes = (kernel_sizes,) * n
        if isinstance(level_layers, int):
            level_layers = (level_layers,) * n
        assert len(kernel_sizes) == n, f'kernel_sizes ({len(kernel_sizes)}) must be of size {n}'
        assert len(level_layers) == n, f'level_layers ({len(level_layers)}) must be of size {n}'
        self.level_layers = level_layers
        self.levels = len(level_layers)
        self.layer_params = []
        feat_channels = feat_channels[::-1]

        prev_channels = 0
        for level in range(self.levels):
            curr_ngf = feat_channels[level]
            prev_channels += curr_ngf
            curr_layers = []
            k = kernel_sizes[level]
            for layer in range(self.level_layers[level]):
                if (not with_out_fc) and level == self.levels - 1 and layer == self.level_layers[level] - 1:
                    curr_ngf = num_classes
                if k > 1:
                    curr_layers.append(HyperPatchInvertedResidual(
                        prev_channels + 2, curr_ngf, k, expand_ratio=expand_ratio, norm_layer=norm_layer,
                        act_layer=act_layer))
                else:
                    curr_layers.append(make_meta_patch_conv2d_block(prev_channels + 2, curr_ngf, k))
                prev_channels = curr_ngf
            self.add_module(f'level_{level}', MetaSequential(*curr_layers))

        if with_out_fc:
            out_fc_layers = [nn.Dropout2d(dropout, True)] if dropout is not None else []
            out_fc_layers.append(
                MetaPatchConv2d(prev_channels, num_classes, out_kernel_size, padding=out_kernel_size // 2))
            self.out_fc = MetaSequential(*out_fc_layers)
        else:
            self.out_fc = None

        self.hyper_params = 0
        self._ranges = [0]
        self.param_groups = []
        for level in range(self.levels):
            lp = getattr(self, f'level_{level}').hyper_params
            self.hyper_params += lp
            self._ranges.append(self.hyper_params)
            self.param_groups.append(lp)
        if with_out_fc:
            self.hyper_params += self.out_fc.hyper_params
            self.param_groups.append(self.out_fc.hyper_params)
        self._ranges.append(self.hyper_params)

    def forward(self, x, w):
        assert isinstance(w, (list, tuple))
        assert len(x) <= self.levels
        p = None
        for level in range(len(x)):
            stage = HF.StageInput(x[-level - 1], p, coords=True)      # cat(coords, skip, bilinear(p)), never built
            p = getattr(self, f'level_{level}')(stage, w[level])
        if self.out_fc is not None:
            p = self.out_fc(p, w[-1])
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
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.bias = bias
        self._ranges = [0]
        parts = divide_feature_legacy(in_channels, out_channels, min_unit)
        for i, out_nc in enumerate(out_channels):
            self._ranges.append(self._ranges[-1] + int(parts[i]))
            self.add_module(f'conv_{i}', nn.Conv2d(int(parts[i]), int(out_nc), kernel_size, stride, padding, dilation,
                                                   groups, bias, padding_mode))

    def forward(self, x):
        return [getattr(self, f'conv_{i}')(x[:, self._ranges[i]:self._ranges[i + 1]])
                for i in range(len(self.out_channels))]

    def extra_repr(self):
        return f'in_channels={self.in_channels}, out_channels={self.out_channels}, bias={self.bias}'


class WeightMapper(nn.Module):
    """v0_1 context head (hyperseg_v0_1.py:249-329): stride-2 2x2 convs down, optional global average at the bottom,
    nearest 2x up + concat + 1x1 "flat" merges, then :class:`Conv2dMulti` emits one weight tensor per level."""

    def __init__(self, in_channels, out_channels, levels=2, bias=False, min_unit=8, down_groups=1, flat_groups=1,
                 weight_groups=1, avg_pool=False):
        super(WeightMapper, self).__init__()
        assert levels > 0, 'levels must be greater than zero'
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.levels = levels
        self.bias = bias
        self.avg_pool = avg_pool
        self.down_groups = down_groups
        self.flat_groups = flat_groups
        self.weight_groups = weight_groups
        min_unit = max(min_unit, weight_groups)
        for level in range(self.levels - 1):
            self.add_module(f'down_{level}', nn.Sequential(
                nn.Conv2d(in_channels, in_channels, kernel_size=2, stride=2, bias=bias, groups=down_groups),
                nn.BatchNorm2d(in_channels), nn.ReLU(inplace=True)))
            self.add_module(f'up_{level}', nn.UpsamplingNearest2d(scale_factor=2))
            flat = [nn.Conv2d(in_channels * 2, in_channels, kernel_size=1, bias=bias, groups=flat_groups),
                    nn.BatchNorm2d(in_channels)]
            if level > 0:
                flat.append(nn.ReLU(inplace=True))
            self.add_module(f'flat_{level}', nn.Sequential(*flat))
        padded = [next_multiply(c, weight_groups) for c in out_channels]
        self.out_conv = Conv2dMulti(in_channels, padded, 1, bias=bias, min_unit=min_unit, groups=weight_groups)

    def forward(self, x):
        if self.levels <= 1:
            return self.out_conv(x)
        feat = [x]
        for level in range(self.levels - 1):
            feat.append(getattr(self, f'down_{level}')(feat[-1]))
        if self.avg_pool and feat[-1].shape[-2:] != (1, 1):
            feat[-1] = F.adaptive_avg_pool2d(feat[-1], 1).expand_as(feat[-1])
        for level in range(self.levels - 2, -1, -1):
            up = getattr(self, f'up_{level}')(feat.pop())
            feat[-1] = getattr(self, f'flat_{level}')(torch.cat((feat[-1], up), dim=1))
        w = self.out_conv(feat[-1])
        if self.weight_groups > 1:
            w = [wi[:, :oc] for wi, oc in zip(w, self.out_channels)]
        return w

    def extra_repr(self):
        return f'in_channels={self.in_channels}, out_channels={self.out_channels}, bias={self.bias}'


class HyperGen(nn.Module):
    """hyperseg_v0_1.py:11-88."""

    def __init__(self, backbone, weight_mapper, in_nc=3, num_classes=3, kernel_sizes=3, level_layers=1, expand_ratio=1,
                 groups=1, inference_hflip=False, inference_gather='mean', with_out_fc=False, decoder_dropout=None):
        super(HyperGen, self).__init__()
        self.inference_hflip = inference_hflip
        self.inference_gather = inference_gather
        self.backbone = backbone()
        feat_channels = [in_nc] + self.backbone.feat_channels[:-1]
        self.decoder = MultiScaleDecoder(feat_channels, 3, num_classes, kernel_sizes, level_layers,
                                         with_out_fc=with_out_fc, out_kernel_size=1, expand_ratio=expand_ratio,
                                         dropout=decoder_dropout)
        self.weight_mapper = weight_mapper(self.backbone.feat_channels[-1], self.decoder.param_groups)

    @property
    def hyper_params(self):
        return self.decoder.hyper_params

    def process_single_tensor(self, x, hflip=False):
        x = torch.flip(x, [-1]) if hflip else x
        features = self.backbone(x)
        weights = self.weight_mapper(features[-1])
        y = self.decoder([t.contiguous() for t in [x] + features[:-1]], weights)
        return torch.flip(y, [-1]) if hflip else y

    def gather_results(self, x, y=None):
        assert x is not None
        if y is None:
            return x
        return (x + y) * 0.5 if self.inference_gather == 'mean' else torch.max(x, y)

    def forward(self, x):
        assert isinstance(x, (list, tuple, torch.Tensor)), 'x must be of type list, tuple, or tensor'
        if isinstance(x, torch.Tensor):
            return self.process_single_tensor(x)
        out_res = x[0].shape[2:]
        out = None
        for p in x:
            if self.inference_hflip:
                p = torch.max(self.process_single_tensor(p), self.process_single_tensor(p, hflip=True))
            else:
                p = self.process_single_tensor(p)
            if p.shape[2:] != out_res:
                p = HF.upsample_bilinear(p.contiguous(), out_res)
            out = self.gather_results(p, out)
        return out


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
