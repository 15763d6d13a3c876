"""MetaSequential -- drop-in for hyperseg/models/layers/meta_sequential.py:5-40.

Same container semantics (children with ``hyper_params`` consume a channel range of ``w`` or the
next entry of a weight list; plain children get ``module(x)``), plus one thing the reference
cannot do: when a dynamic convolution is followed by an eval-mode BatchNorm2d and/or ReLU/ReLU6
(the blocks built by make_*_patch_conv2d_block), the three are executed as ONE HIP launch through
the child's ``forward_fused`` (BN folded to scale/shift, activation in the epilogue).
The channel slice is passed as a view -- the reference's ``.contiguous()`` copy (line 35) is gone.
"""
from itertools import accumulate

import torch
import torch.nn as nn

from ... import functional as HF


def x_is_cuda(x):
    return (x.skip if isinstance(x, HF.StageInput) else x).is_cuda


def _act_code(m):
    if isinstance(m, nn.ReLU6):
        return HF.ACT_RELU6
    if isinstance(m, nn.ReLU):
        return HF.ACT_RELU
    return None


class MetaSequential(nn.Sequential):
    def __init__(self, *args):
        super().__init__(*args)
        # child i consumes channels [_ranges[i], _ranges[i + 1]) of a weight tensor (or the next entry of a weight list);
        # the container advertises the total, so containers nest (the reference's contract, meta_sequential.py:10-17)
        counts = [getattr(m, 'hyper_params', 0) for m in self]
        self._ranges = [0] + list(accumulate(counts))
        self.hyper_params = self._ranges[-1]
        self._folded = {}

    def train(self, mode=True):
        HF.bump_weights_epoch()             # a mode switch: parameters / BN statistics may have moved without a version bump
        return super().train(mode)

    def _fold(self, idx, bn):
        cache = self._folded.get(idx)
        if cache is None:
            cache = self._folded.setdefault(idx, HF.FoldedBN())     # atomic: replica threads share this dict (nn.DataParallel)
        return cache.get(bn)

    def forward(self, x, w):
        mods = list(self)
        n = len(mods)
        w_count = 0
        i = 0
        while i < n:
            module = mods[i]
            if self._ranges[i] < self._ranges[i + 1]:
                if isinstance(w, (list, tuple)):
                    wi = w[w_count]
                else:
                    wi = w[:, self._ranges[i]:self._ranges[i + 1]]     # clamped view, no copy
                w_count += 1
                if hasattr(module, 'forward_fused'):
                    j = i + 1
                    scale = shift = None
                    act = HF.ACT_NONE
                    grad = torch.is_grad_enabled() and mods[j].weight is not None and mods[j].weight.requires_grad \
                        if j < n and isinstance(mods[j], nn.BatchNorm2d) else False
                    if j < n and isinstance(mods[j], nn.BatchNorm2d) and not mods[j].training and not grad:
                        scale, shift = self._fold(j, mods[j])
                        j += 1
                    elif j < n and isinstance(mods[j], nn.BatchNorm2d) and x_is_cuda(x):
                        # training: the convolution through autograd, then BatchNorm (batch statistics) + activation as the fused
                        # training kernels (autograd.bn_act: two launches per direction instead of MIOpen's BN + a clamp + their adjoints)
                        from ... import autograd as HA
                        bn, j = mods[j], j + 1
                        act_layer = None
                        if j < n and _act_code(mods[j]) is not None:
                            act_layer, j = mods[j], j + 1
                        x = HA.bn_act(bn, act_layer, module.forward_fused(x, wi, None, None, HF.ACT_NONE))
                        i = j
                        continue
                    if j < n and _act_code(mods[j]) is not None:
                        act = _act_code(mods[j])
                        j += 1
                    x = module.forward_fused(x, wi, scale, shift, act)
                    i = j
                    continue
                x = module(x, wi)
            else:
                if isinstance(x, HF.StageInput):
                    x = x.materialize()
                x = module(x)
            i += 1
        return x
