"""Inference-time preparation of the STOCK PyTorch parts of a HyperGen model (encoder + context head).

Not part of the decoder hot path and not custom kernels: standard, mathematically equivalent transforms of
eval-mode modules that remove launches from the frame (the encoder is ~90 % of the frame):
  * fold every eval BatchNorm2d that directly follows a convolution into that convolution (w' = w * g/sqrt(v+eps),
    b' = beta - mean * g/sqrt(v+eps)); the BN becomes nn.Identity;
  * optionally switch the encoder to channels_last.
The decoder modules are left untouched (their BatchNorms are folded inside the HIP kernels' epilogues).
The state dict changes (BN entries disappear), so apply it AFTER loading a checkpoint.
"""
import torch
import torch.nn as nn


@torch.no_grad()
def _fold(conv, bn):
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    shift = bn.bias - bn.running_mean * scale
    conv.weight.mul_(scale.view(-1, 1, 1, 1))
    if conv.bias is not None:
        conv.bias.mul_(scale).add_(shift)
    else:
        conv.bias = nn.Parameter(shift.clone())


def _fold_pairs(module, pairs):
    n = 0
    for conv_name, bn_name in pairs:
        conv, bn = getattr(module, conv_name, None), getattr(module, bn_name, None)
        if isinstance(conv, nn.Conv2d) and isinstance(bn, nn.BatchNorm2d) and not bn.training:
            _fold(conv, bn)
            setattr(module, bn_name, nn.Identity())
            n += 1
    return n


def _fold_sequential(seq):
    n = 0
    mods = list(seq)
    for i in range(len(mods) - 1):
        if isinstance(mods[i], nn.Conv2d) and isinstance(mods[i + 1], nn.BatchNorm2d) and not mods[i + 1].training:
            _fold(mods[i], mods[i + 1])
            seq[i + 1] = nn.Identity()
            n += 1
    return n


class FusedDepthwiseBNSwish(nn.Module):
    """depthwise conv + eval BatchNorm + swish of one MBConv block as ONE ``hs_depthwise_conv_fwd`` launch
    (MIOpen has no tuned fp32 depthwise solver on ROCm 7.2: Winograd-per-group / naive kernels, ~half of the frame).
    Holds the folded BN affine as non-persistent buffers; the filter stays the block's own ``_depthwise_conv.weight``."""

    def __init__(self, conv, bn):
        super().__init__()
        with torch.no_grad():
            scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            shift = bn.bias - bn.running_mean * scale
        self.register_buffer('scale', scale.detach().clone(), persistent=False)
        self.register_buffer('shift', shift.detach().clone(), persistent=False)
        self._conv = [conv]                      # not registered twice
        self._exp_t = None                       # (Csq, C) transposed SE expand weight, built on first use
        self.k, self.stride = conv.kernel_size[0], conv.stride[0]
        if conv._pad is not None:                # asymmetric TF-"SAME": (left, right, top, bottom)
            self.pad_l, self.pad_t = conv._pad[0], conv._pad[2]
            self.pad_w, self.pad_h = conv._pad[0] + conv._pad[1], conv._pad[2] + conv._pad[3]
        else:
            self.pad_t, self.pad_l = conv.padding
            self.pad_h, self.pad_w = 2 * conv.padding[0], 2 * conv.padding[1]

    def forward(self, x, blk):
        """x -> project_conv(SE(swish(bn1(depthwise(x))))) of MBConv block ``blk`` (before its bn2)."""
        import torch.nn.functional as F
        from .. import functional as HF
        b = x.shape[0]
        h, w = x.shape[-2:]
        ho = (h + self.pad_h - self.k) // self.stride + 1
        wo = (w + self.pad_w - self.k) // self.stride + 1
        y, partial = HF.depthwise_conv_bn_act(x.contiguous(), self._conv[0].weight, self.stride, self.pad_t, self.pad_l,
                                              (ho, wo), self.scale, self.shift, act=3, pool=True)
        red, exp, proj = blk._se_reduce, blk._se_expand, blk._project_conv
        if self._exp_t is None or self._exp_t.device != x.device:
            self._exp_t = exp.weight.detach().flatten(1).t().contiguous()
        if b == 1:
            # gate folded into the 1x1 project weights: no elementwise pass over the activation
            wp = HF.se_gate(partial, 1, ho * wo, red.weight, red.bias, self._exp_t, exp.bias, w_proj=proj.weight)
            return F.conv2d(y, wp[0])
        gate = HF.se_gate(partial, b, ho * wo, red.weight, red.bias, self._exp_t, exp.bias)
        return proj(y * gate[:, :, None, None])


def prepare_for_inference(model, fold_bn=True, channels_last=False, fused_depthwise=False):
    """In-place; returns the number of BatchNorms folded.  ``model`` is a HyperGen in eval mode.
    ``fused_depthwise`` swaps each MBConv block's depthwise conv + BN + swish for the fused HIP kernel (do it BEFORE
    ``fold_bn`` touches those BatchNorms: handled here)."""
    assert not model.training, 'call model.eval() first'
    folded = 0
    if fused_depthwise:
        for blk in model.backbone._blocks:
            conv = blk._depthwise_conv
            if conv.kernel_size[0] in (3, 5) and conv.stride[0] in (1, 2) and isinstance(blk._bn1, nn.BatchNorm2d):
                blk._fused_dw = FusedDepthwiseBNSwish(conv, blk._bn1)
    if fold_bn:
        bb = model.backbone
        folded += _fold_pairs(bb, [('_conv_stem', '_bn0'), ('_conv_head', '_bn1')])
        for blk in bb._blocks:
            pairs = [('_expand_conv', '_bn0'), ('_project_conv', '_bn2')]
            if blk._fused_dw is None:
                pairs.append(('_depthwise_conv', '_bn1'))
            folded += _fold_pairs(blk, pairs)
        for m in list(bb.children()) + list(model.weight_mapper.modules()):
            if isinstance(m, nn.Sequential):
                folded += _fold_sequential(m)
    if channels_last:
        model.backbone.to(memory_format=torch.channels_last)
        model.weight_mapper.to(memory_format=torch.channels_last)
    return folded
