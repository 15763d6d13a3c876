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


def prepare_for_inference(model, fold_bn=True, channels_last=False):
    """In-place; returns the number of BatchNorms folded.  ``model`` is a HyperGen in eval mode."""
    assert not model.training, 'call model.eval() first'
    folded = 0
    if fold_bn:
        bb = model.backbone
        folded += _fold_pairs(bb, [('_conv_stem', '_bn0'), ('_conv_head', '_bn1')])
        for blk in bb._blocks:
            folded += _fold_pairs(blk, [('_expand_conv', '_bn0'), ('_depthwise_conv', '_bn1'), ('_project_conv', '_bn2')])
        for m in list(bb.children()) + list(model.weight_mapper.modules()):
            if isinstance(m, nn.Sequential):
                folded += _fold_sequential(m)
    if channels_last:
        model.backbone.to(memory_format=torch.channels_last)
        model.weight_mapper.to(memory_format=torch.channels_last)
    return folded
