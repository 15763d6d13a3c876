"""MetaConv2d -- drop-in for hyperseg/models/layers/meta_conv.py:9-230 on the HIP path.

The reference folds the batch into conv groups (``F.conv2d(groups=B*g)``, lines 163-186).  Here a
per-sample dynamic convolution is simply the patch-wise kernel with a 1x1 weight grid: the (B, hp)
weight matrix already IS a patch-major bank, so there is no re-layout at all.
"""
import torch
import torch.nn as nn
from torch.nn.modules.utils import _pair

from ... import functional as HF
from .meta_sequential import MetaSequential


def _require_inference(*tensors):
    if torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors):
        raise NotImplementedError('hyperseg_amd: this entry point is the fused inference path; gradients flow through '
                                  'hyperseg_amd.autograd (module forward in training mode)')


PADDING_MODES = ('zeros', 'reflect', 'replicate', 'circular')


def check_padding_mode(mode):
    """Same ValueError the reference raises (meta_conv.py:149-151, meta_patch.py:21-24)."""
    if mode not in PADDING_MODES:
        raise ValueError(f"padding_mode must be one of {set(PADDING_MODES)}, but got padding_mode='{mode}'")
    return mode


def assemble_block(conv, out_nc, norm_layer, act_layer, dropout):
    """conv -> norm -> activation -> dropout, each optional, as one MetaSequential (the layout of the reference's
    ``make_meta_*_block`` factories; MetaSequential fuses conv + eval-BN + ReLU/ReLU6 into one launch)."""
    if not (dropout is None or isinstance(dropout, float)):
        raise AssertionError('dropout must be None or a float')
    tail = [norm_layer(out_nc) if norm_layer is not None else None, act_layer,
            nn.Dropout(dropout) if dropout is not None else None]
    return MetaSequential(conv, *[m for m in tail if m is not None])


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
        for name, nc in (('in_channels', in_channels), ('out_channels', out_channels)):
            if nc % groups:
                raise ValueError(f'{name} must be divisible by groups')
        self.padding_mode = check_padding_mode(padding_mode)
        self.in_channels, self.out_channels, self.groups = in_channels, out_channels, groups
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        # rows of the weight vector one sample needs: Cout x Cin/groups x kh x kw
        self.hyper_params = out_channels * (in_channels // groups) * self.kernel_size[0] * self.kernel_size[1]

    def _check_supported(self):
        """(k, padding) of a square, stride-1, dilation-1 conv -- what the PATCH-wise wrappers (MetaPatch, HyperPatch) can
        hand to the patch-convolution kernels; anything else raises there, as the patch semantics of a strided conv are
        not something any reference configuration defines."""
        kh, kw = self.kernel_size
        if kh != kw or self.stride != (1, 1) or self.dilation != (1, 1) or self.padding[0] != self.padding[1]:
            raise NotImplementedError('hyperseg_amd patch-wise kernels cover square kernels, stride 1, dilation 1 '
                                      '(everything the reference configs instantiate)')
        return kh, self.padding[0]

    def _same_padded(self):
        """Square kernel, stride 1, dilation 1, padding (k - 1) / 2: what every reference configuration instantiates and
        what the LDS-tiled patch-convolution kernels (forward and backward) cover."""
        kh, kw = self.kernel_size
        return kh == kw and self.stride == (1, 1) and self.dilation == (1, 1) and \
            self.padding[0] == self.padding[1] and 2 * self.padding[0] == kh - 1

    def forward_fused(self, x, w, scale=None, shift=None, act=HF.ACT_NONE):
        assert x.shape[0] == w.shape[0]
        if w.dim() != 2 or w.shape[1] != self.hyper_params:
            raise ValueError(f'w must be (B, {self.hyper_params}), got {tuple(w.shape)}')
        from ... import autograd as HA
        if not self._same_padded():
            # the rest of the reference's argument set (meta_conv.py:141-186): the general kernel; under autograd its own
            # Function (hs_meta_conv_fwd + hs_meta_conv_bwd; no reference config trains one, the class is differentiable anyway)
            if HA.needs_grad(x if isinstance(x, torch.Tensor) else x.skip, w):
                xt = HA.materialize_stage(x) if isinstance(x, HF.StageInput) else x
                y = HA.meta_conv_general(xt, w, self.out_channels, self.kernel_size, self.stride, self.padding, self.dilation,
                                         self.padding_mode, self.groups)
                return _apply_epilogue(y, scale, shift, act)
            return HF.meta_conv(x, w if w.stride(1) == 1 else w.contiguous(), self.out_channels, self.kernel_size, self.stride,
                                self.padding, self.dilation, self.padding_mode, self.groups, scale, shift, act)
        k, pad = self.kernel_size[0], self.padding[0]
        if HA.needs_grad(x if isinstance(x, torch.Tensor) else x.skip, w):
            xt = HA.materialize_stage(x) if isinstance(x, HF.StageInput) else x
            y = HA.patch_conv_apply(xt, w, (1, 1), self.out_channels, k, pad, self.padding_mode, self.groups)
            return _apply_epilogue(y, scale, shift, act)
        if w.stride(1) != 1 and w.shape[1] != 1:
            w = w.contiguous()
        return HF.patch_conv(x, (1, 1), w, self.out_channels, k, pad, self.padding_mode, self.groups,
                             scale, shift, act)

    def forward(self, x, w):
        return self.forward_fused(x, w)

    def extra_repr(self):
        parts = [str(self.in_channels), str(self.out_channels), f'kernel_size={self.kernel_size}', f'stride={self.stride}']
        for name, default in (('padding', (0, 0)), ('dilation', (1, 1)), ('groups', 1), ('padding_mode', 'zeros')):
            if getattr(self, name) != default:
                parts.append(f'{name}={getattr(self, name)}')
        return ', '.join(parts)


def make_meta_conv2d_block(in_nc, out_nc, kernel_size=3, stride=1, padding=None, dilation=1, groups=1,
                           padding_mode='reflect', norm_layer=nn.BatchNorm2d, act_layer=nn.ReLU(True), dropout=None):
    """[MetaConv2d, norm, act, Dropout?] in a MetaSequential (meta_conv.py:202-230)."""
    pad = kernel_size // 2 if padding is None else padding
    conv = MetaConv2d(in_nc, out_nc, kernel_size, stride, pad, dilation, groups, padding_mode)
    return assemble_block(conv, out_nc, norm_layer, act_layer, dropout)
