"""EfficientNet encoder with per-resolution feature taps -- stock PyTorch-ROCm (no custom kernels).

BASELINE config 2 keeps the encoder on stock PyTorch; this file exists so that the HyperSeg models
are complete end-to-end and reference checkpoints load: parameter names, shapes and arithmetic
follow hyperseg/models/backbones/efficientnet.py (the vendored lukemelas EfficientNet plus HyperSeg's
additions: ``_res_feat_mask`` taps :176-204, ``_feat_fc_i`` 1x1 conv + BN channel reducers scaled by
``out_feat_scale`` :206-222, ``extract_features_list`` :319-363).  Written from the architecture
description, not from that file:

* MBConv = [1x1 expand + BN + swish] -> depthwise kxk (TF "SAME" padding) + BN + swish ->
  squeeze-excite (ratio of the block's INPUT filters) -> 1x1 project + BN (+ identity skip);
* width/depth scaling by (w, d) with the divisor-8 rounding rule; BN momentum 0.01, eps 1e-3;
* "SAME" padding is STATIC: computed once from the model's nominal resolution (240 for B1, 300 for B3)
  as the reference does (efficientnet_utils.py:247-274), not from the actual input size -- checkpoints
  were trained that way, so it is reproduced.
"""
import math
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

# (expand, kernel, stride, in, out, repeats) of the B0 stages; squeeze-excite ratio 0.25 everywhere
_B0_STAGES = [
    (1, 3, 1, 32, 16, 1),
    (6, 3, 2, 16, 24, 2),
    (6, 5, 2, 24, 40, 2),
    (6, 3, 2, 40, 80, 3),
    (6, 5, 1, 80, 112, 3),
    (6, 5, 2, 112, 192, 4),
    (6, 3, 1, 192, 320, 1),
]
# name -> (width, depth, nominal resolution, dropout)
_SCALING = {
    'efficientnet-b0': (1.0, 1.0, 224, 0.2), 'efficientnet-b1': (1.0, 1.1, 240, 0.2),
    'efficientnet-b2': (1.1, 1.2, 260, 0.3), 'efficientnet-b3': (1.2, 1.4, 300, 0.3),
    'efficientnet-b4': (1.4, 1.8, 380, 0.4), 'efficientnet-b5': (1.6, 2.2, 456, 0.4),
    'efficientnet-b6': (1.8, 2.6, 528, 0.5), 'efficientnet-b7': (2.0, 3.1, 600, 0.5),
}
_BN_MOMENTUM, _BN_EPS, _SE_RATIO, _DIVISOR = 0.01, 1e-3, 0.25, 8


def round_filters(filters, width):
    if not width:
        return filters
    filters *= width
    new = max(_DIVISOR, int(filters + _DIVISOR / 2) // _DIVISOR * _DIVISOR)
    if new < 0.9 * filters:          # never round down by more than 10 %
        new += _DIVISOR
    return int(new)


def round_repeats(repeats, depth):
    return int(math.ceil(depth * repeats)) if depth else repeats


def _out_size(size, stride):
    return None if size is None else int(math.ceil(size / stride))


class SamePadConv2d(nn.Conv2d):
    """Conv2d with TensorFlow 'SAME' zero padding fixed at construction for a nominal square input of
    ``image_size`` pixels (asymmetric: the extra pixel goes right/bottom)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, image_size=None, groups=1, bias=True):
        super().__init__(in_channels, out_channels, kernel_size, stride, 0, 1, groups, bias)
        k, s = self.kernel_size[0], self.stride[0]
        total = max((int(math.ceil(image_size / s)) - 1) * s + k - image_size, 0)
        self._pad = (total // 2, total - total // 2, total // 2, total - total // 2) if total > 0 else None
        if self._pad is not None and total % 2 == 0:
            # symmetric: let the convolution pad (one kernel instead of pad + conv)
            self.padding = (total // 2, total // 2)
            self._pad = None

    def forward(self, x):
        if self._pad is not None:
            x = F.pad(x, self._pad)
        return self._conv_forward(x, self.weight, self.bias)


def _inference_only(module, x):
    """CUDA tensor, eval mode, and no gradient can be asked of this call."""
    if not x.is_cuda or module.training:
        return False
    if not torch.is_grad_enabled():
        return True
    return not (x.requires_grad or any(p.requires_grad for p in module.parameters()))


class MBConvBlock(nn.Module):
    def __init__(self, in_f, out_f, expand, kernel, stride, image_size):
        super().__init__()
        self.stride, self.in_f, self.out_f, self.expand = stride, in_f, out_f, expand
        mid = in_f * expand
        bn = partial(nn.BatchNorm2d, momentum=_BN_MOMENTUM, eps=_BN_EPS)
        if expand != 1:
            self._expand_conv = SamePadConv2d(in_f, mid, 1, image_size=image_size, bias=False)
            self._bn0 = bn(mid)
        self._depthwise_conv = SamePadConv2d(mid, mid, kernel, stride, image_size=image_size, groups=mid, bias=False)
        self._bn1 = bn(mid)
        squeezed = max(1, int(in_f * _SE_RATIO))
        self._se_reduce = SamePadConv2d(mid, squeezed, 1, image_size=1)
        self._se_expand = SamePadConv2d(squeezed, mid, 1, image_size=1)
        self._project_conv = SamePadConv2d(mid, out_f, 1, image_size=_out_size(image_size, stride), bias=False)
        self._bn2 = bn(out_f)
        self._fused_dw = None       # set by utils.inference.prepare_for_inference(fused_depthwise=True)
        self._fused_active = None   # per forward, set by EfficientNet.extract_features_list: all blocks fused or none
                                    # (deferred BN shifts chain from block to block)

    def forward(self, inputs, drop_connect_rate=None):
        active = self._fused_active if self._fused_active is not None else _inference_only(self, inputs)
        if self._fused_dw is not None and active:
            return self._fused_dw(inputs, self)         # the whole block in 4 HIP launches (utils/inference.py)
        x = inputs
        if self.expand != 1:
            x = F.silu(self._bn0(self._expand_conv(x)))
        x = F.silu(self._bn1(self._depthwise_conv(x)))
        gate = self._se_expand(F.silu(self._se_reduce(F.adaptive_avg_pool2d(x, 1))))
        x = self._bn2(self._project_conv(torch.sigmoid(gate) * x))
        if self.stride == 1 and self.in_f == self.out_f:
            if drop_connect_rate and self.training:      # stochastic depth
                keep = 1.0 - drop_connect_rate
                mask = torch.floor(keep + torch.rand(x.shape[0], 1, 1, 1, dtype=x.dtype, device=x.device))
                x = x / keep * mask
            x = x + inputs
        return x


class EfficientNet(nn.Module):
    """``forward(x)`` returns, with ``return_features=True``, the list of the LAST feature map of every
    resolution (strides 2, 4, 8, 16, 32; each optionally reduced by its ``_feat_fc_i``) followed by the
    swish(BN(conv_head)) map at stride 32; ``feat_channels`` lists their channel counts."""

    def __init__(self, model_name, out_feat_scale=0.25, head=None, return_features=False, pool=False,
                 num_classes=1000, drop_connect_rate=0.2, in_channels=3):
        super().__init__()
        if model_name not in _SCALING:
            raise ValueError(f'unknown model {model_name}; choose from {sorted(_SCALING)}')
        width, depth, res, dropout = _SCALING[model_name]
        self.return_features, self.pool, self.drop_connect_rate = return_features, pool, drop_connect_rate
        self.out_feat_scale = out_feat_scale
        bn = partial(nn.BatchNorm2d, momentum=_BN_MOMENTUM, eps=_BN_EPS)

        size = res
        stem = round_filters(32, width)
        self._conv_stem = SamePadConv2d(in_channels, stem, 3, 2, image_size=size, bias=False)
        self._bn0 = bn(stem)
        size = _out_size(size, 2)

        self._blocks = nn.ModuleList()
        self._res_feat_mask, feat_nc = [], []
        for expand, kernel, stride, in_f, out_f, repeats in _B0_STAGES:
            in_f, out_f = round_filters(in_f, width), round_filters(out_f, width)
            repeats = round_repeats(repeats, depth)
            if stride > 1:
                self._res_feat_mask[-1] = True      # the block before a stride ends a resolution
            self._res_feat_mask += [False] * repeats
            feat_nc += [out_f] * repeats
            for r in range(repeats):
                self._blocks.append(MBConvBlock(in_f if r == 0 else out_f, out_f, expand, kernel,
                                                stride if r == 0 else 1, size))
                if r == 0:
                    size = _out_size(size, stride)
        self._res_feat_mask[-1] = True
        self.feat_channels = [c for c, m in zip(feat_nc, self._res_feat_mask) if m]

        if out_feat_scale is not None:
            for i, in_nc in enumerate(self.feat_channels):
                scale = out_feat_scale[i] if isinstance(out_feat_scale, (list, tuple)) else out_feat_scale
                out_nc = int(round(in_nc * scale))
                if scale != 1.:
                    self.add_module(f'_feat_fc_{i}', nn.Sequential(
                        SamePadConv2d(in_nc, out_nc, 1, image_size=res, bias=False), bn(out_nc)))
                else:
                    setattr(self, f'_feat_fc_{i}', None)
                self.feat_channels[i] = out_nc

        head_nc = round_filters(1280, width)
        self.feat_channels.append(head_nc)
        self._conv_head = SamePadConv2d(feat_nc[-1], head_nc, 1, image_size=size, bias=False)
        self._bn1 = bn(head_nc)
        self._avg_pooling = nn.AdaptiveAvgPool2d(1)
        self._dropout = nn.Dropout(dropout)
        self._fc = head(head_nc, num_classes) if head is not None else None
        self._fused_head, self._fused_fc, self._fused_stem = None, None, None      # set by utils.inference.prepare_for_inference

    def _fused_ok(self, inputs):
        """The prepared (fused HIP) route is taken for a whole forward or not at all -- its blocks hand deferred BN shifts
        to one another: CUDA, eval mode, nothing that needs a gradient (the kernels have no backward: an eval-mode
        backbone under autograd, e.g. a frozen-BN fine-tune, takes the stock route), and every stage's H*W a multiple of
        4 (16-byte rows in hs_affine_act_fwd / the library-GEMM epilogues; odd pyramid scales fall back)."""
        if self._fused_stem is None and self._fused_head is None and all(b._fused_dw is None for b in self._blocks):
            return False
        if not _inference_only(self, inputs):
            return False
        h, w = (inputs.shape[2] + 1) // 2, (inputs.shape[3] + 1) // 2           # stem: stride 2, TF-"SAME"
        for blk in self._blocks:
            st = blk._depthwise_conv.stride[0]
            if (h * w) % 4 != 0:
                return False
            h, w = (h + st - 1) // st, (w + st - 1) // st
        return (h * w) % 4 == 0

    def extract_features_list(self, inputs):
        use = self._fused_ok(inputs)
        for blk in self._blocks:
            blk._fused_active = use
        try:
            return self._extract_features_list(inputs, use)
        finally:
            for blk in self._blocks:
                blk._fused_active = None

    def _extract_features_list(self, inputs, use):
        f0 = self._blocks[0]._fused_dw if len(self._blocks) else None
        if self._fused_stem is not None and use and f0 is not None and getattr(f0, '_stem', None) is not None:
            x = inputs                          # block 0's fused route runs the stem itself (one launch with its depthwise half)
        elif self._fused_stem is not None and use:
            x = self._fused_stem(inputs)
        else:
            x = F.silu(self._bn0(self._conv_stem(inputs)))
        feats = []
        n = len(self._blocks)
        for idx, block in enumerate(self._blocks):
            rate = self.drop_connect_rate * float(idx) / n if self.drop_connect_rate else None
            x = block(x, drop_connect_rate=rate)
            if self._res_feat_mask[idx]:
                fc = getattr(self, f'_feat_fc_{len(feats)}', None) if self.out_feat_scale is not None else None
                fused = self._fused_fc is not None and use and str(len(feats)) in self._fused_fc
                feats.append(self._fused_fc[str(len(feats))](x) if fused else (x if fc is None else fc(x)))
        if self._fused_head is not None and use:
            x = self._fused_head(x)
        else:
            x = F.silu(self._bn1(self._conv_head(x)))
        if self.pool:
            x = self._avg_pooling(x).flatten(1)
        x = self._dropout(x)
        if self._fc is not None:
            x = self._fc(x)
        feats.append(x)
        return feats

    def forward(self, inputs):
        feats = self.extract_features_list(inputs)
        return feats if self.return_features else feats[-1]


def efficientnet(model_name, pretrained=False, head=nn.Linear, **kwargs):
    """Factory with the reference's signature (efficientnet.py:493-502)."""
    if pretrained:
        raise RuntimeError('pretrained ImageNet weights need network access; load a checkpoint with '
                           'load_state_dict instead (parameter names match the reference)')
    return EfficientNet(model_name, head=head, **kwargs)
