"""MetaConv2d -- drop-in for hyperseg/models/layers/meta_conv.py:9-230 on the HIP path.

The reference folds the batch into conv groups (``F.conv2d(groups=B*g)``, lines 163-186).  Here a
per-sample dynamic convolution is simply the patch-wise kernel with a 1x1 weight grid: the (B, hp)
weight matrix already IS a patch-major bank, so there is no re-layout at all.
"""
import numpy as np
import torch
import torch.nn as nn
from torch.nn.modules.utils import _pair

from ... import functional as HF
from .meta_sequential import MetaSequential


def _require_inference(*tensors):
    if torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors):
        raise NotImplementedError('hyperseg_amd: this entry point is the fused inference path; gradients flow through '
                                  'hyperseg_amd.autograd (module forward in training mode)')


def _apply_epilogue(y, scale, shift, act):
    """Epilogue with stock ops (training path: the autograd kernels have no fused epilogue)."""
    if scale is not None:
        y = y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if act == HF.ACT_RELU:
        y = torch.relu(y)
    elif act == HF.ACT_RELU6:
        y = torch.clamp(y, 0.0, 6.0)
    return y


class MetaConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 padding_mode='zeros'):
        super(MetaConv2d, self).__init__()
        if in_channels % groups != 0:
            raise ValueError('in_channels must be divisible by groups')
        if out_channels % groups != 0:
            raise ValueError('out_channels must be divisible by groups')
        valid_padding_modes = {'zeros', 'reflect', 'replicate', 'circular'}
        if padding_mode not in valid_padding_modes:
            raise ValueError(
                f"padding_mode must be one of {valid_padding_modes}, but got padding_mode='{padding_mode}'")
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = _pair(kernel_size)
        self.stride = _pair(stride)
        self.padding = _pair(padding)
        self.dilation = _pair(dilation)
        self.groups = groups
        self.padding_mode = padding_mode
        self.hyper_params = int(np.prod((out_channels, in_channels // groups) + self.kernel_size))

    def _check_supported(self):
        kh, kw = self.kernel_size
        if kh != kw or self.stride != (1, 1) or self.dilation != (1, 1) or self.padding[0] != self.padding[1]:
            raise NotImplementedError('hyperseg_amd kernels cover square kernels, stride 1, dilation 1 '
                                      '(everything the reference configs instantiate)')
        return kh, self.padding[0]

    def forward_fused(self, x, w, scale=None, shift=None, act=HF.ACT_NONE):
        k, pad = self._check_supported()
        assert x.shape[0] == w.shape[0]
        if w.dim() != 2 or w.shape[1] != self.hyper_params:
            raise ValueError(f'w must be (B, {self.hyper_params}), got {tuple(w.shape)}')
        from ... import autograd as HA
        if HA.needs_grad(x if isinstance(x, torch.Tensor) else x.skip, w):
            xt = HA.materialize_stage(x) if isinstance(x, HF.StageInput) else x
            y = HA.PatchConv.apply(xt, w, (1, 1), self.out_channels, k, pad, self.padding_mode, self.groups)
            return _apply_epilogue(y, scale, shift, act)
        if w.stride(1) != 1 and w.shape[1] != 1:
            w = w.contiguous()
        return HF.patch_conv(x, (1, 1), w, self.out_channels, k, pad, self.padding_mode, self.groups,
                             scale, shift, act)

    def forward(self, x, w):
        return self.forward_fused(x, w)

    def extra_repr(self):
        s = ('{in_channels}, {out_channels}, kernel_size={kernel_size}'
             ', stride={stride}')
        if self.padding != (0,) * len(self.padding):
            s += ', padding={padding}'
        if self.dilation != (1,) * len(self.dilation):
            s += ', dilation={dilation}'
        if self.groups != 1:
            s += ', groups={groups}'
        if self.padding_mode != 'zeros':
            s += ', padding_mode={padding_mode}'
        return s.format(**self.__dict__)


def make_meta_conv2d_block(in_nc, out_nc, kernel_size=3, stride=1, padding=None, dilation=1, groups=1,
                           padding_mode='reflect', norm_layer=nn.BatchNorm2d, act_layer=nn.ReLU(True), dropout=None):
    """[MetaConv2d, norm, act, Dropout?] in a MetaSequential (meta_conv.py:202-230)."""
    assert dropout is None or isinstance(dropout, float)
    padding = kernel_size // 2 if padding is None else padding
    layers = [MetaConv2d(in_nc, out_nc, kernel_size, stride, padding, dilation, groups, padding_mode)]
    if norm_layer is not None:
        layers.append(norm_layer(out_nc))
    if act_layer is not None:
        layers.append(act_layer)
    if dropout is not None:
        layers.append(nn.Dropout(dropout))
    return MetaSequential(*layers)
