"""MetaPatch / MetaPatchConv2d -- drop-in for hyperseg/models/layers/meta_patch.py:9-257.

The reference pads the whole image, unfolds it into halo tiles, folds the batch of tiles into conv
groups and folds the result back (lines 35-57: six memory passes around one small conv).  Here:
one ``hs_bank_pack_fwd`` launch turns the (B, hp, fh, fw) weight tensor into a patch-major bank and
one ``hs_patch_conv_fwd`` launch does padding + gather + conv (+ BN + activation when called from a
MetaSequential) with the tile living in LDS.
"""
import torch.nn as nn
from torch.nn.modules.utils import _pair

from ... import functional as HF
from .meta_conv import MetaConv2d, _apply_epilogue, assemble_block, check_padding_mode


class MetaPatch(nn.Module):
    """Dynamic patch-wise wrapper.  Only a wrapped :class:`MetaConv2d` has a HIP kernel; any other
    wrapped module raises (the reference's generic unfold/fold route is not reproduced)."""

    def __init__(self, module: nn.Module, padding=0, padding_mode='reflect'):
        super(MetaPatch, self).__init__()
        self.hyper_module = module
        self.padding, self.padding_mode = _pair(padding), check_padding_mode(padding_mode)

    @property
    def hyper_params(self):
        return self.hyper_module.hyper_params

    def forward_fused(self, x, weight, scale=None, shift=None, act=HF.ACT_NONE):
        conv = self.hyper_module
        if not isinstance(conv, MetaConv2d):
            raise NotImplementedError('hyperseg_amd.MetaPatch has a HIP kernel only for a wrapped MetaConv2d')
        k, inner_pad = conv._check_supported()
        if inner_pad != 0 or self.padding[0] != self.padding[1]:
            raise NotImplementedError('MetaPatch expects the wrapped conv to be unpadded (meta_patch.py:190-193)')
        if weight.dim() != 4 or weight.shape[1] < conv.hyper_params:
            raise ValueError(f'weight must be (B, >={conv.hyper_params}, fh, fw), got {tuple(weight.shape)}')
        fh, fw = weight.shape[-2:]
        h, w = x.shape[-2:]
        if h % fh != 0 or w % fw != 0:
            raise ValueError(f'input {h}x{w} does not tile into the {fh}x{fw} weight grid')
        from ... import autograd as HA
        probe = [x.skip, x.prev] if isinstance(x, HF.StageInput) else [x]
        if HA.needs_grad(weight, *probe):
            # training: HIP forward + HIP backward kernels through autograd, stock glue around them
            xt = HA.materialize_stage(x) if isinstance(x, HF.StageInput) else x
            y = HA.patch_conv_train(xt, weight, conv.out_channels, k, self.padding[0], self.padding_mode, conv.groups,
                                    conv.hyper_params)
            return _apply_epilogue(y, scale, shift, act)
        # a bank the context head already wrote patch-major (HF.BankRef) is consumed as is
        bank = weight.bank if isinstance(weight, HF.BankRef) else HF.bank_pack(weight, 0, conv.hyper_params)
        return HF.patch_conv(x, (fh, fw), bank, conv.out_channels, k, self.padding[0], self.padding_mode,
                             conv.groups, scale, shift, act)

    def forward(self, x, weight):
        return self.forward_fused(x, weight)


class MetaPatchConv2d(MetaPatch):
    """Patch-wise dynamic 2D convolution: every cell of the (fh, fw) weight grid has its own filters.
    forward(x (B,C,H,W), weight (B,hp,fh,fw)) -> (B,Cout,H,W)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 padding_mode='reflect'):
        conv = MetaConv2d(in_channels, out_channels, kernel_size, stride, 0, dilation, groups)
        super(MetaPatchConv2d, self).__init__(conv, padding, padding_mode)

    # the wrapped conv's attributes, exposed the way the reference's subclass stores them
    in_channels = property(lambda self: self.hyper_module.in_channels)
    out_channels = property(lambda self: self.hyper_module.out_channels)
    kernel_size = property(lambda self: self.hyper_module.kernel_size)
    groups = property(lambda self: self.hyper_module.groups)

    def __repr__(self):
        m = self.hyper_module
        parts = [str(m.in_channels), str(m.out_channels), f'kernel_size={m.kernel_size}', f'stride={m.stride}']
        if self.padding != (0, 0):
            parts.append(f'padding={self.padding}')
        if m.groups != 1:
            parts.append(f'groups={m.groups}')
        if self.padding_mode != 'zeros':
            parts.append(f'padding_mode={self.padding_mode}')
        return f"{type(self).__name__}({', '.join(parts)})"


def make_meta_patch_conv2d_block(in_nc, out_nc, kernel_size=3, stride=1, padding=None, dilation=1, groups=1,
                                 padding_mode='reflect', norm_layer=nn.BatchNorm2d, act_layer=nn.ReLU(True),
                                 dropout=None):
    """[MetaPatchConv2d, norm, act, Dropout?] in a MetaSequential (meta_patch.py:228-257)."""
    pad = kernel_size // 2 if padding is None else padding
    conv = MetaPatchConv2d(in_nc, out_nc, kernel_size, stride, pad, dilation, groups, padding_mode)
    return assemble_block(conv, out_nc, norm_layer, act_layer, dropout)
