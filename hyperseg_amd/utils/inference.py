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


def _bn_affine(bn):
    with torch.no_grad():
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        shift = bn.bias - bn.running_mean * scale
    return scale.detach().clone(), shift.detach().clone()


class FusedPointwise(nn.Module):
    """1x1 conv + eval BatchNorm (+ activation) as one ``hs_pointwise_conv_fwd`` launch.  The conv stays where it is in
    the model (weights are read through a reference); only the folded BN affine lives here (non-persistent buffers)."""

    def __init__(self, conv, bn, act=0):
        super().__init__()
        assert conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.groups == 1 and conv.bias is None
        scale, shift = _bn_affine(bn)
        self.register_buffer('scale', scale, persistent=False)
        self.register_buffer('shift', shift, persistent=False)
        self._conv = [conv]
        self.act = act

    def uses_mfma(self, x):
        """Small-K, many-pixel layers run as one fused MFMA GEMM; large-K layers keep the stock (rocBLAS) GEMM."""
        return x.shape[1] <= 96 and x.shape[2] * x.shape[3] >= 8192

    def forward(self, x, gate=None, residual=None, w_scaled=None):
        """``gate`` (B, Cin): SE gate applied to the input (MFMA path), or ``w_scaled`` (Cout, Cin, 1, 1): the conv
        weights with the gate already folded in (stock-GEMM path, batch 1).  The stock GEMM is followed by ONE fused
        BatchNorm + activation + skip-add launch."""
        import torch.nn.functional as F
        from .. import functional as HF
        conv = self._conv[0]
        x = x.contiguous()
        b, cin, h, w = x.shape
        if self.uses_mfma(x) and w_scaled is None:
            return HF.pointwise_conv(x, conv.weight, gate, self.scale, self.shift, self.act, residual)
        if (h * w) % 4 != 0:
            raise NotImplementedError('feature maps with H*W % 4 != 0')
        if w_scaled is not None:
            y = F.conv2d(x, w_scaled)
        elif gate is None:
            y = F.conv2d(x, conv.weight)
        else:
            y = F.conv2d(x * gate[:, :, None, None], conv.weight)
        return HF.affine_act_(y, self.scale, self.shift, self.act, residual)


class FusedMBConv(nn.Module):
    """A whole MBConv block in 4 HIP launches: [1x1 expand + BN + swish] -> [depthwise + BN + swish + SE pooling] ->
    [SE gate] -> [gate * 1x1 project + BN + skip add]  (stock: 15 launches; MIOpen has no tuned fp32 depthwise solver on
    ROCm 7.2 -- Winograd-per-group / naive kernels cost half of the frame).  Filters stay the block's own parameters."""

    def __init__(self, blk):
        super().__init__()
        conv = blk._depthwise_conv
        self.expand = FusedPointwise(blk._expand_conv, blk._bn0, act=3) if blk.expand != 1 else None
        self.project = FusedPointwise(blk._project_conv, blk._bn2, act=0)
        scale, shift = _bn_affine(blk._bn1)
        self.register_buffer('scale', scale, persistent=False)
        self.register_buffer('shift', shift, persistent=False)
        self.k, self.stride = conv.kernel_size[0], conv.stride[0]
        if conv._pad is not None:                # asymmetric TF-"SAME": (left, right, top, bottom)
            self.pad_l, self.pad_t = conv._pad[0], conv._pad[2]
            self.pad_w, self.pad_h = conv._pad[0] + conv._pad[1], conv._pad[2] + conv._pad[3]
        else:
            self.pad_t, self.pad_l = conv.padding
            self.pad_h, self.pad_w = 2 * conv.padding[0], 2 * conv.padding[1]
        self.skip = blk.stride == 1 and blk.in_f == blk.out_f
        self._exp_t = None                       # (Csq, C) transposed SE expand weight, built on first use

    def forward(self, inputs, blk):
        from .. import functional as HF
        x = inputs.contiguous()
        if self.expand is not None:
            x = self.expand(x)
        b, _, h, w = x.shape
        ho = (h + self.pad_h - self.k) // self.stride + 1
        wo = (w + self.pad_w - self.k) // self.stride + 1
        y, partial = HF.depthwise_conv_bn_act(x, blk._depthwise_conv.weight, self.stride, self.pad_t, self.pad_l,
                                              (ho, wo), self.scale, self.shift, act=3, pool=True)
        red, exp = blk._se_reduce, blk._se_expand
        if self._exp_t is None or self._exp_t.device != x.device:
            self._exp_t = exp.weight.detach().flatten(1).t().contiguous()
        skip = inputs.contiguous() if self.skip else None
        if b == 1 and not self.project.uses_mfma(y):
            # large K: gate folded into the project weights (tiny kernel) instead of an elementwise pass over y
            wp = HF.se_gate(partial, 1, ho * wo, red.weight, red.bias, self._exp_t, exp.bias, w_proj=blk._project_conv.weight)
            return self.project(y, residual=skip, w_scaled=wp[0])
        gate = HF.se_gate(partial, b, ho * wo, red.weight, red.bias, self._exp_t, exp.bias)
        return self.project(y, gate=gate, residual=skip)


def prepare_for_inference(model, fold_bn=True, channels_last=False, fused_depthwise=False):
    # fused_depthwise: historical name -- it now fuses every MBConv block end to end (4 launches), the head and the
    # feature reducers
    """In-place; returns the number of BatchNorms folded.  ``model`` is a HyperGen in eval mode.
    ``fused_depthwise`` swaps each MBConv block's depthwise conv + BN + swish for the fused HIP kernel (do it BEFORE
    ``fold_bn`` touches those BatchNorms: handled here)."""
    assert not model.training, 'call model.eval() first'
    folded = 0
    if fused_depthwise:
        bb = model.backbone
        for blk in bb._blocks:
            conv = blk._depthwise_conv
            if conv.kernel_size[0] in (3, 5) and conv.stride[0] in (1, 2) and isinstance(blk._bn1, nn.BatchNorm2d):
                blk._fused_dw = FusedMBConv(blk)
        bb._fused_head = FusedPointwise(bb._conv_head, bb._bn1, act=3)
        bb._fused_fc = nn.ModuleDict()
        for i in range(len(bb.feat_channels) - 1):
            fc = getattr(bb, f'_feat_fc_{i}', None)
            if isinstance(fc, nn.Sequential):
                bb._fused_fc[str(i)] = FusedPointwise(fc[0], fc[1], act=0)
    if fold_bn:
        bb = model.backbone
        folded += _fold_pairs(bb, [('_conv_stem', '_bn0')] + ([] if getattr(bb, '_fused_head', None) is not None else [('_conv_head', '_bn1')]))
        for blk in bb._blocks:
            if blk._fused_dw is None:
                folded += _fold_pairs(blk, [('_expand_conv', '_bn0'), ('_project_conv', '_bn2'), ('_depthwise_conv', '_bn1')])
        fused_fc = getattr(bb, '_fused_fc', None)
        for name, m in list(bb.named_children()) + list(model.weight_mapper.named_modules()):
            if isinstance(m, nn.Sequential) and not (fused_fc is not None and name.startswith('_feat_fc_')):
                folded += _fold_sequential(m)
    if channels_last:
        model.backbone.to(memory_format=torch.channels_last)
        model.weight_mapper.to(memory_format=torch.channels_last)
    return folded
