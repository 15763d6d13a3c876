"""Pieces shared by the three HyperSeg model variants of this package (v1_0, v1_0_unify, v0_1): the HyperGen wrapper
logic (single tensor / pyramid + h-flip inference), per-level argument normalisation, coordinate buffers."""
import numbers

import torch
import torch.nn as nn

from .. import functional as HF


def per_level(value, n, name):
    """Broadcast a scalar hyper-parameter to ``n`` levels, or check a sequence's length."""
    if isinstance(value, numbers.Number):
        return (value,) * n
    if len(value) != n:
        raise AssertionError(f'{name} ({len(value)}) must be of size {n}')
    return tuple(value)


def plan_levels(feat_channels, level_channels, kernel_sizes, level_layers, expand_ratio, groups, num_classes,
                with_out_fc):
    """Channel bookkeeping of a v1_0-style decoder, separated from module construction: for every level (coarse -> fine)
    the list of its layers as dicts ``{cin, cout, k, expand, groups}``.  Rules (hyperseg_v1_0.py:139-163): a level's first
    layer sees everything carried up from the coarser level plus this level's skip feature plus two coordinate channels;
    a level emits ``level_channels[l]`` (or its skip width) channels, except that the very last layer of the decoder emits
    the class logits unless a separate output layer follows."""
    skips = list(feat_channels)[::-1]
    n = len(skips) if level_channels is None else len(level_channels)
    plan, carried = [], 0
    for lvl in range(n):
        width = skips[lvl] if level_channels is None else level_channels[lvl]
        carried += skips[lvl]
        layers = []
        for j in range(level_layers[lvl]):
            last_of_decoder = lvl == n - 1 and j == level_layers[lvl] - 1
            cout = num_classes if (last_of_decoder and not with_out_fc) else width
            layers.append(dict(cin=carried + 2, cout=cout, k=kernel_sizes[lvl], expand=expand_ratio[lvl],
                               groups=groups[lvl] if isinstance(groups, (list, tuple)) else groups))
            carried = cout
        plan.append(layers)
    return plan, carried


def coordinate_grid(h, w, device=None):
    """(1, 2, h, w): channel 0 = x in [-1, 1] over W, channel 1 = y over H, endpoints inclusive.  Same values as the
    reference's cached buffers; the HIP kernels regenerate them analytically."""
    xs = torch.linspace(-1, 1, steps=w, device=device).view(1, w).expand(h, w)
    ys = torch.linspace(-1, 1, steps=h, device=device).view(h, 1).expand(h, w)
    return torch.stack([xs, ys], dim=0).unsqueeze(0).contiguous()


def register_coordinate_buffers(module, coords_res, levels):
    """``coord{h}_{w}`` buffers for every resolution of every listed pyramid -- kept only so that reference checkpoints
    load with strict=True."""
    for res in coords_res or ():
        for i in range(levels):
            h, w = res[0] // 2 ** i, res[1] // 2 ** i
            module.register_buffer(f'coord{h}_{w}', coordinate_grid(h, w))


class EpochOnModeSwitch:
    """Mixin (before nn.Module in the bases): every train() / eval() switch invalidates the parameter-derived caches of the
    inference route (functional.bump_weights_epoch) -- training steps change parameters and BatchNorm statistics through
    paths that do not bump tensor versions (graph replays, raw-pointer kernels)."""

    def train(self, mode=True):
        HF.bump_weights_epoch()
        return super().train(mode)


class HyperGenBase(EpochOnModeSwitch, nn.Module):
    """backbone -> context head -> dynamic decoder, with the reference's list-input (image pyramid) and horizontal-flip
    inference modes (hyperseg_v1_0.py:52-91).  Subclasses create ``backbone``, ``decoder`` and ``weight_mapper``."""

    inference_hflip = False
    inference_gather = 'mean'

    @property
    def hyper_params(self):
        return self.decoder.hyper_params

    def process_single_tensor(self, x, hflip=False, masks=False):
        if hflip:
            x = torch.flip(x, [-1])
        features = self.backbone(x)
        head_out = self.weight_mapper(features[-1])
        if isinstance(head_out, torch.Tensor):
            head_out = head_out.contiguous()
        pyramid = [t.contiguous() for t in [x] + features[:-1]]
        y = self.decoder(pyramid, head_out, masks=True) if masks else self.decoder(pyramid, head_out)
        return torch.flip(y, [-1]) if hflip else y

    @torch.no_grad()
    def segment(self, x):
        """uint8 class masks (B, H, W) == ``self(x).argmax(1)`` (the reference's test.py:171 / test_fps.py:194 epilogue).
        For a single tensor in eval mode the argmax is taken inside the final upsample kernel and the full-resolution
        logits are never written; pyramid / h-flip inference falls back to the logits path."""
        if isinstance(x, torch.Tensor) and not self.training and not self.inference_hflip:
            return self.process_single_tensor(x, masks=True)
        return self(x).argmax(1).to(torch.uint8)

    def gather_results(self, x, y=None):
        assert x is not None
        if y is None:
            return x
        return (x + y) * 0.5 if self.inference_gather == 'mean' else torch.max(x, y)

    def forward(self, x):
        if isinstance(x, torch.Tensor):
            return self.process_single_tensor(x)
        assert isinstance(x, (list, tuple)), 'x must be of type list, tuple, or tensor'
        out_res = x[0].shape[2:]          # the first pyramid level sets the output resolution
        merged = None
        for level in x:
            y = self.process_single_tensor(level)
            if self.inference_hflip:
                y = torch.max(y, self.process_single_tensor(level, hflip=True))
            if y.shape[2:] != out_res:
                y = HF.upsample_bilinear(y.contiguous(), out_res)
            merged = self.gather_results(y, merged)
        return merged
