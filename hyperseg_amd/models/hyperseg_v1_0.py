"""HyperSeg v1.0 on the MI355X decoder path -- drop-in for hyperseg/models/hyperseg_v1_0.py.

Same class names, constructor arguments, ``forward(x, s)`` signatures, ``hyper_params`` bookkeeping
and state-dict keys as the reference (HyperSeg-M Cityscapes and HyperSeg-S/L CamVid configs), so
reference checkpoints load with ``strict=True``.  What differs is what runs:

* the encoder (:mod:`.backbones.efficientnet`) and the context head (:class:`WeightMapper`) are
  stock PyTorch-ROCm;
* every decoder level is ONE fused HIP launch (plus one bank-producing launch): the stage input
  ``cat(coords, skip, bilinear2x(prev))`` is generated inside the kernel's prologue, the per-patch
  filter bank is produced patch-major by ``hs_signal2weights_fwd`` and consumed immediately, and
  BatchNorm + ReLU/ReLU6 live in the epilogue (reference: hyperseg_v1_0.py:221-253, 328-376,
  486-498: ~11-21 ATen kernels per level).

Quirks kept on purpose: ``signal_index`` is 0 for every level (D-1), the
signal reaches the modules through MetaSequential's clamped slice (D-2), ``weight_groups`` lists are
consumed by ``pop(0)`` (D-3), coordinate buffers exist for checkpoint compatibility but the values
are generated analytically (D-11).
"""
from functools import partial
from itertools import groupby

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.modules.utils import _pair

from .. import functional as HF
from .. import autograd as HA
from ._common import EpochOnModeSwitch, HyperGenBase, coordinate_grid, per_level, plan_levels, register_coordinate_buffers
from .layers.meta_conv import MetaConv2d, _apply_epilogue, _require_inference, assemble_block, check_padding_mode
from .layers.meta_sequential import MetaSequential


def next_multiply(x, base):
    return type(x)(np.ceil(x / base) * base)


class _SignalToWeights:
    """Mixin for the modules that own a ``signal2weights`` grouped 1x1 conv (hyperseg_v1_0.py:315-326,
    473-484, 529-541).  ``_bank(s, rows)`` runs the conv as one HIP launch that writes the
    patch-major bank directly; ``apply_signal2weights`` keeps the reference's tensor-returning form."""

    def _init_s2w_state(self):
        self.signal_channels = None
        self.signal_index = None
        self.signal2weights = None
        self._s2w_t = HF.TransposedS2W()

    def _make_signal2weights(self, signal_channels, signal_index, groups, weight_channels):
        self.signal_channels = int(signal_channels)
        self.signal_index = int(signal_index)
        self.signal2weights = nn.Conv2d(int(signal_channels), int(weight_channels), 1, bias=False, groups=int(groups))

    def _s2w_layer(self, rows):
        """Descriptor of this module's signal2weights for the decoder-wide single launch."""
        conv = self.signal2weights
        return dict(wsw_t=self._s2w_t.get(conv), signal_index=self.signal_index,
                    signal_channels=self.signal_channels, groups=conv.groups, rows=rows)

    def _bank(self, s, rows):
        if isinstance(s, HF.BankRef):
            if s.rows != rows:
                raise ValueError(f'bank has {s.rows} rows, module needs {rows}')
            return s.bank
        conv = self.signal2weights
        if conv is None:
            # no hypernetwork head: ``s`` already holds the weights (B, hp, fh, fw)
            return HF.bank_pack(s, 0, rows)
        _require_inference(s, conv.weight)
        return HF.signal2weights(s, self._s2w_t.get(conv), self.signal_index, self.signal_channels,
                                 conv.groups, rows)

    def _weights_train(self, s):
        """Differentiable reference-layout weights (B, >= hp, fh, fw): the grouped 1x1 conv as a stock op (training).  The rows past hp
        (``next_multiply`` padding of the grouped conv, hyperseg_v1_0.py:473-477) are NOT sliced off on the GPU route: every consumer
        takes its row count explicitly (autograd.BankPack / patch_conv_train), and a slice would cost a zero fill + a strided copy in
        every backward (its adjoint pads the gradient back to the conv's width)."""
        hp = int(self.hyper_params)
        if isinstance(s, HF.BankRef):
            raise RuntimeError('a precomputed bank cannot carry gradients')
        if self.signal2weights is None:
            return s[:, :hp]
        conv = self.signal2weights
        sig = s[:, self.signal_index:self.signal_index + self.signal_channels]
        if s.is_cuda and conv.bias is None:
            # The grouped 1x1 conv as ONE strided-batched GEMM per direction (rocBLAS / hipBLASLt, forward and both gradients
            # through autograd): MIOpen runs this shape -- a few hundred pixels, up to 64 groups -- on its naive direct kernel
            # (naive_conv_ab_nonpacked_*: 0.3 ms of the config-5 step, and far worse under bf16 autocast, which is why bf16
            # training was SLOWER than fp32 in round 2, profiles/round2_train_step_kernels.txt).  Same sums, same order of
            # operands; hyperseg_v1_0.py:479-484.
            b, _, fh, fw = sig.shape
            g = conv.groups
            wc, k = conv.weight.shape[0], conv.weight.shape[1]
            w = conv.weight.view(g, wc // g, k)
            x = sig.reshape(b, g, k, fh * fw)
            return torch.einsum('grk,bgkp->bgrp', w, x).reshape(b, wc, fh, fw)
        return conv(sig)[:, :hp]

    def _train_mode(self, x, s):
        if isinstance(s, HF.TrainBank):
            # a TrainBank only fits the autograd.* Functions (PatchConv.apply & co. run under no_grad as well); the inference kernels
            # take tensors / BankRefs -- so ``decoder.train()`` inside ``torch.no_grad()`` stays on this route
            return True
        probe = [x.skip, x.prev] if isinstance(x, HF.StageInput) else [x]
        params = [self.signal2weights.weight] if self.signal2weights is not None else []
        return not isinstance(s, HF.BankRef) and HA.needs_grad(s, *probe, *params)

    def apply_signal2weights(self, s):
        """(B, hp, fh, fw) weights in the reference's channel-major layout (diagnostics / API parity)."""
        if self.signal2weights is None:
            return s
        b, _, fh, fw = s.shape
        hp = int(self.hyper_params)
        bank = self._bank(s, hp)
        return bank[:, :hp].reshape(b, fh, fw, hp).permute(0, 3, 1, 2)


class HyperPatchNoPadding(EpochOnModeSwitch, nn.Module, _SignalToWeights):
    """k=1 dynamic patch-wise conv fed by the signal (hyperseg_v1_0.py:455-498) -> Op A."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dilation=1, groups=1):
        super().__init__()
        for name, nc in (('in_channels', in_channels), ('out_channels', out_channels)):
            if nc % groups:
                raise ValueError(f'{name} must be divisible by groups')       # the reference's messages (:458-461)
        self.in_channels, self.out_channels, self.groups = in_channels, out_channels, groups
        self.kernel_size, self.stride, self.dilation = _pair(kernel_size), _pair(stride), _pair(dilation)
        kh, kw = self.kernel_size
        self.hyper_params = out_channels * (in_channels // groups) * kh * kw   # filter-bank rows of one patch
        self._init_s2w_state()

    def init_signal2weights(self, signal_channels, signal_index=0, groups=1):
        self._make_signal2weights(signal_channels, signal_index, groups, next_multiply(self.hyper_params, groups))

    def s2w_layer(self, device):
        return self._s2w_layer(self.hyper_params)

    def forward_fused(self, x, s, scale=None, shift=None, act=HF.ACT_NONE):
        if self.kernel_size != (1, 1) or self.stride != (1, 1) or self.dilation != (1, 1):
            raise NotImplementedError('HyperPatchNoPadding: only the k=1, stride 1 form the reference builds '
                                      '(padding == 0 <=> kernel_size == 1, hyperseg_v1_0.py:748-750)')
        if isinstance(s, HF.SignalRef):
            # the bank is generated inside the consumer; if the kernel does not cover the shape, materialise it after all
            y = HF.patch_conv_gen(x, s, self.out_channels, scale, shift, act) if self.groups == 1 else None
            if y is not None:
                return y
            s = HF.signal2weights_multi(s.signal, [s.layer])[0]
        if self._train_mode(x, s):
            xt = HA.materialize_stage(x) if isinstance(x, HF.StageInput) else x
            if isinstance(s, HF.TrainBank):          # the bank already exists, patch-major and differentiable (autograd.S2WBanksTrain)
                y = HA.patch_conv_apply(xt, s.bank, s.grid, self.out_channels, 1, 0, 'zeros', self.groups)
            else:
                y = HA.patch_conv_train(xt, self._weights_train(s), self.out_channels, 1, 0, 'zeros', self.groups,
                                        self.hyper_params)
            return _apply_epilogue(y, scale, shift, act)
        fh, fw = s.shape[-2:]
        bank = self._bank(s, self.hyper_params)
        return HF.patch_conv(x, (fh, fw), bank, self.out_channels, 1, 0, 'zeros', self.groups, scale, shift, act)

    def forward(self, x, s):
        return self.forward_fused(x, s)


class HyperPatch(EpochOnModeSwitch, nn.Module, _SignalToWeights):
    """Dynamic patch-wise block with image-level padding fed by the signal (hyperseg_v1_0.py:501-557)."""

    def __init__(self, module: nn.Module, padding=0, padding_mode='reflect'):
        super().__init__()
        self.padding_mode = check_padding_mode(padding_mode)                 # same ValueError text as the reference (:506-510)
        self.hyper_module, self.padding = module, _pair(padding)
        self._init_s2w_state()

    @property
    def hyper_params(self):
        return self.hyper_module.hyper_params

    def init_signal2weights(self, signal_channels, signal_index=0, groups=1):
        # NB: no next_multiply here in the reference (hyperseg_v1_0.py:532-535)
        self._make_signal2weights(signal_channels, signal_index, groups, self.hyper_params)

    def s2w_layer(self, device):
        return self._s2w_layer(self.hyper_params)

    def forward_fused(self, x, s, scale=None, shift=None, act=HF.ACT_NONE):
        conv = self.hyper_module
        if not isinstance(conv, MetaConv2d):
            raise NotImplementedError('hyperseg_amd.HyperPatch has a HIP kernel only for a wrapped MetaConv2d')
        k, inner_pad = conv._check_supported()
        if inner_pad != 0 or self.padding[0] != self.padding[1]:
            raise NotImplementedError('HyperPatch expects the wrapped conv to be unpadded')
        if self._train_mode(x, s):
            xt = HA.materialize_stage(x) if isinstance(x, HF.StageInput) else x
            if isinstance(s, HF.TrainBank):
                y = HA.patch_conv_apply(xt, s.bank, s.grid, conv.out_channels, k, self.padding[0], self.padding_mode, conv.groups)
                return _apply_epilogue(y, scale, shift, act)
            y = HA.patch_conv_train(xt, self._weights_train(s), conv.out_channels, k, self.padding[0],
                                    self.padding_mode, conv.groups, conv.hyper_params)
            return _apply_epilogue(y, scale, shift, act)
        fh, fw = s.shape[-2:]
        bank = self._bank(s, conv.hyper_params)
        return HF.patch_conv(x, (fh, fw), bank, conv.out_channels, k, self.padding[0], self.padding_mode,
                             conv.groups, scale, shift, act)

    def forward(self, x, s):
        return self.forward_fused(x, s)


class HyperPatchConv2d(HyperPatch):
    """forward(x (B,C,H,W), s (B,Cs,fh,fw)) -> (B,Cout,H,W) (hyperseg_v1_0.py:560-725)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 padding_mode='reflect'):
        conv = MetaConv2d(in_channels, out_channels, kernel_size, stride, 0, dilation, groups)
        super(HyperPatchConv2d, self).__init__(conv, padding, padding_mode)

    @property
    def in_channels(self):
        return self.hyper_module.in_channels

    @property
    def out_channels(self):
        return self.hyper_module.out_channels

    @property
    def kernel_size(self):
        return self.hyper_module.kernel_size

    @property
    def groups(self):
        return self.hyper_module.groups


class HyperPatchInvertedResidual(EpochOnModeSwitch, nn.Module, _SignalToWeights):
    """Per-patch MobileNetV2 block on a reflect halo tile (hyperseg_v1_0.py:281-376) -> Op C, one launch."""

    def __init__(self, in_nc, out_nc, kernel_size=3, stride=1, expand_ratio=1, norm_layer=nn.BatchNorm2d,
                 act_layer=nn.ReLU6(inplace=True), padding_mode='reflect'):
        super().__init__()
        assert stride in (1, 2)
        hidden = int(round(in_nc * expand_ratio))
        self.in_nc, self.out_nc, self.hidden_dim, self.stride = in_nc, out_nc, hidden, stride
        self.kernel_size, self.padding, self.padding_mode = _pair(kernel_size), (1, 1), padding_mode
        self.use_res_connect = stride == 1 and in_nc == out_nc
        self.act_layer = act_layer
        # state-dict keys bn1 / bn2 / bn3 (Appendix C)
        self.bn1, self.bn2, self.bn3 = norm_layer(hidden), norm_layer(hidden), norm_layer(out_nc)
        # rows of the patch's filter bank: pw1 [hidden x in_nc] | depthwise [hidden x kh x kw] | pw3 [out_nc x hidden]
        kh, kw = self.kernel_size
        sizes = (in_nc * hidden, hidden * kh * kw, hidden * out_nc)
        self._ranges = [0, sizes[0], sizes[0] + sizes[1], sum(sizes)]
        self.hyper_params = self._ranges[-1]
        self._init_s2w_state()
        self._folded = [HF.FoldedBN(), HF.FoldedBN(), HF.FoldedBN()]
        self._unit_affine = {}

    def init_signal2weights(self, signal_channels, signal_index=0, groups=1):
        self._make_signal2weights(signal_channels, signal_index, groups, next_multiply(self.hyper_params, groups))

    @staticmethod
    def _is_identity(m):
        """A normalisation slot emptied by the FPS harness' BN -> identity switch (test_fps.py:147, 319-332)."""
        return isinstance(m, nn.Identity) or type(m).__name__ == 'Unit'

    def _affine_of(self, idx, bn, dev):
        if self._is_identity(bn):
            n = self.out_nc if idx == 2 else self.hidden_dim
            key = (idx, n, dev)                          # one entry per device: replicas on other GPUs share this dict by reference
            ent = self._unit_affine.get(key)
            if ent is None:
                ent = (torch.ones(n, device=dev), torch.zeros(n, device=dev))
                HF.publish_ready(torch.device(dev))
                self._unit_affine[key] = ent
            return ent
        return self._folded[idx].get(bn)

    def _check_supported(self):
        if self.kernel_size != (3, 3) or self.stride != 1 or self.padding_mode != 'reflect' or \
                not isinstance(self.act_layer, nn.ReLU6) or \
                not all(isinstance(b, nn.BatchNorm2d) or self._is_identity(b) for b in (self.bn1, self.bn2, self.bn3)):
            raise NotImplementedError('hs_patch_ir_fwd implements the block every reference config builds: '
                                      '3x3 depthwise, stride 1, reflect halo, BatchNorm2d, ReLU6')

    def s2w_layer(self, device):
        return self._s2w_layer(self.hyper_params)

    def _run_train(self, x, s, residual):
        """Training / gradient path (hyperseg_v1_0.py:328-376 semantics incl. BN1 statistics over the duplicated halo
        pixels, Appendix D-4): the halo tiles are laid side by side as one "tiled image", on which pw1 is a k=1 patch
        conv, the depthwise 3x3 a zero-padded k=3 patch conv whose tile interiors are kept, pw3 a k=1 patch conv on the
        re-assembled image -- three HIP convolutions with HIP backward kernels; BatchNorm / ReLU6 / gather are stock."""
        xt = HA.materialize_stage(x) if isinstance(x, HF.StageInput) else x
        b, c, h, wd = xt.shape
        r1, r2, r3 = self._ranges[1], self._ranges[2], self._ranges[3]
        if isinstance(s, HF.TrainBank):              # patch-major and differentiable already (autograd.S2WBanksTrain)
            bank, (fh, fw) = s.bank, s.grid
        else:
            w = self._weights_train(s)
            fh, fw = w.shape[-2:]
            bank = HA.BankPack.apply(w, r3)
        ph, pw = h // fh, wd // fw
        grid = (fh, fw)
        own = HA.tiles_supported(xt)
        # patch-major tiles (B fh fw, C, ph+2, pw+2): every operand of a patch one contiguous run; the 1x1 layer on them is a patch convolution
        # with a (1, 1) grid over B fh fw frames -- same kernels, same bank rows -- and BatchNorm sees the same multiset of values per channel
        pm = own and HA.USE_PATCH_MAJOR_TILES and HA.USE_HIP_DW_TILES and h % fh == 0 and wd % fw == 0 and pw % 2 == 0 \
            and b * fh * fw * max(c, self.hidden_dim) <= 65535
        if own:                                                                    # one gather each way (hs_halo_tiles_fwd / _bwd)
            tiled = HA.HaloTiles.apply(xt, grid, pm)
        else:
            xp = F.pad(xt, (1, 1, 1, 1), mode='reflect')
            tiles = xp.unfold(2, ph + 2, ph).unfold(3, pw + 2, pw)                 # B C fh fw ph+2 pw+2
            tiled = tiles.permute(0, 1, 2, 4, 3, 5).reshape(b, c, fh * (ph + 2), fw * (pw + 2))
        bank1, bank2, bank3 = HA.BankSlices.apply(bank, r1, r2, r3)             # one concatenation in the backward instead of 3 x (zeros + copy) + 2 adds
        y = HA.patch_conv_apply(tiled, bank1, (1, 1) if pm else grid, self.hidden_dim, 1, 0, 'zeros', 1)
        if pm:
            # BatchNorm1 + ReLU6 applied to the raw tiles ON LOAD by the depthwise layer (autograd.DwTilesBN): no normalised copy
            y = HA.dw_tiles_bn(self.bn1, self.act_layer, y, bank2, (h, wd), grid, True)
        elif own and HA.dw_tiles_supported(y, (h, wd), grid):
            # the valid depthwise 3x3 of every tile, straight to the (B, hidden, H, W) map: one launch per direction and operand
            y = HA.dw_tiles_bn(self.bn1, self.act_layer, y, bank2, (h, wd), grid, False)
        else:
            y = HA.bn_act(self.bn1, self.act_layer, y)
            if own and HA.tiles_supported(y):
                y = HA.patch_conv_apply(y, bank2, grid, self.hidden_dim, 3, 1, 'zeros', self.hidden_dim)
                y = HA.TileInterior.apply(y, (h, wd), grid)
            else:
                y = HA.patch_conv_apply(y, bank2, grid, self.hidden_dim, 3, 1, 'zeros', self.hidden_dim)
                y = y.reshape(b, self.hidden_dim, fh, ph + 2, fw, pw + 2)[:, :, :, 1:-1, :, 1:-1].reshape(b, self.hidden_dim, h, wd)
        # BatchNorm2 + ReLU6 applied to the raw hidden map ON LOAD by the last 1x1 layer (autograd.PatchConvBN) where it is covered
        y = HA.patch_conv_bn(self.bn2, self.act_layer, y, bank3, grid, self.out_nc)
        y = HA.bn_act(self.bn3, None, y)
        return xt + y if residual else y

    def _run(self, x, s, residual):
        self._check_supported()
        norms = [b for b in (self.bn1, self.bn2, self.bn3) if isinstance(b, nn.BatchNorm2d)]
        if self._train_mode(x, s) or any(b.training for b in norms) or HA.needs_grad(*[b.weight for b in norms]):
            return self._run_train(x, s, residual)
        stage = HF.as_stage(x)
        if stage.channels != self.in_nc:
            raise ValueError(f'expected {self.in_nc} input channels, got {stage.channels}')
        fh, fw = s.shape[-2:]
        bank = self._bank(s, self.hyper_params)
        bns = [self._affine_of(i, bn, bank.device) for i, bn in enumerate((self.bn1, self.bn2, self.bn3))]
        return HF.patch_ir(stage, (fh, fw), bank, self.hidden_dim, self.out_nc, *bns, residual=residual,
                           math=getattr(self, 'ir_math', None))

    def conv(self, x, s):
        return self._run(x, s, False)

    def forward(self, x, s):
        return self._run(x, s, self.use_res_connect)


def make_hyper_patch_conv2d_block(in_nc, out_nc, kernel_size=3, stride=1, padding=None, dilation=1, groups=1,
                                  padding_mode='reflect', norm_layer=nn.BatchNorm2d, act_layer=nn.ReLU(True),
                                  dropout=None):
    """[HyperPatchNoPadding | HyperPatchConv2d, norm, act, Dropout?] (hyperseg_v1_0.py:728-760)."""
    pad = kernel_size // 2 if padding is None else padding
    conv = (HyperPatchNoPadding(in_nc, out_nc, kernel_size, stride, dilation, groups) if pad == 0 else
            HyperPatchConv2d(in_nc, out_nc, kernel_size, stride, pad, dilation, groups, padding_mode))
    return assemble_block(conv, out_nc, norm_layer, act_layer, dropout)


_HYPER_TYPES = (HyperPatchConv2d, HyperPatchNoPadding, HyperPatchInvertedResidual)


def get_hyper_params(model):
    """hyper_params of every signal-fed module, depth first (hyperseg_v1_0.py:256-266)."""
    found = []
    for _, m in model.named_children():
        if isinstance(m, _HYPER_TYPES):
            found.append(m.hyper_params)
        else:
            found.extend(get_hyper_params(m))
    return found


def init_signal2weights(model, signal_features, signal_index=0, weight_groups=1):
    """Hands every signal-fed module its share of signal channels and its group count, depth first.

    Faithful to hyperseg_v1_0.py:269-278 including its quirk: the running ``signal_index`` advances
    only across DIRECT children of one container and is not propagated back out of a recursion, so
    with one hyper-module per level every level reads signal channels starting at 0 (Appendix D-1)."""
    for _, m in model.named_children():
        if isinstance(m, _HYPER_TYPES):
            nc = signal_features.pop(0)
            g = weight_groups.pop(0) if isinstance(weight_groups, list) else weight_groups
            m.init_signal2weights(nc, signal_index, g)
            signal_index += nc
        else:
            init_signal2weights(m, signal_features, signal_index, weight_groups)


def divide_feature(in_feature, out_features, min_unit=8):
    """Split ``in_feature`` channels between consumers in proportion to ``out_features``, in multiples
    of ``min_unit``; equal consumers get equal shares and the group with the smallest total takes the
    remainder.  Must reproduce hyperseg_v1_0.py:763-810 exactly (checkpoint shapes depend on it)."""
    assert in_feature % min_unit == 0, f'in_feature ({in_feature}) must be divisible by min_unit ({min_unit})'
    units = in_feature // min_unit
    order = np.argsort(out_features)
    sorted_vals = np.array(out_features)[order]
    groups = [(val, order[list(idx)]) for val, idx in groupby(range(len(order)), lambda i: sorted_vals[i])]
    groups.sort(key=lambda g: g[0] * len(g[1]), reverse=True)
    ratio = float(units) / sum(out_features)
    share = [len(members) for _, members in groups]          # one unit per consumer to start with
    left = units - sum(share)
    for gi, (val, members) in enumerate(groups):
        if gi == len(groups) - 1:
            share[-1] += left
            break
        n = len(members)
        want = max(val * n * ratio, n)
        want = want // n * n - n                             # float floor-division, as the reference
        want = min(want, left)
        share[gi] += want
        left -= want
        if left == 0:
            break
    out = np.zeros(len(out_features), dtype=int)
    for gi, (_, members) in enumerate(groups):
        for m in members:
            out[m] = share[gi] // len(members) * min_unit
    return out


def run_decoder_chain(owner, seqs, refs, x):
    """The coarse levels of a v1_0 / unify decoder through functional.K1Chain (hs_k1_chain_fwd / hs_decoder_chain_fwd): levels 0-2 when
    each is [HyperPatchNoPadding(k = 1, groups = 1), eval BatchNorm?, ReLU | ReLU6?] on patches of 1, 2 and 4 pixels with a materialised
    bank -- the layout every v1_0 reference configuration builds (hyperseg_v1_0.py:728-760) -- and, when ``owner.chain_ir`` says so, the
    first inverted-residual level behind them.  ``seqs``: the levels' MetaSequentials; ``refs``: their BankRefs; ``x``: the feature
    pyramid.  Returns (output, number of levels done) or None (the caller runs the levels one launch each).  The K1Chain object (it
    owns the launch's workspace) lives on ``owner``."""
    from .layers.meta_sequential import _act_code
    if len(seqs) < 3:
        return None
    skips, bnk, couts, affines, acts = [], [], [], [], []
    for l in range(3):
        seq = seqs[l]
        mods = list(seq)
        while len(mods) == 1 and isinstance(mods[0], MetaSequential):      # level_<l> = MetaSequential(block), block = MetaSequential(conv, norm, act)
            seq = mods[0]
            mods = list(seq)
        ref = refs[l]
        if not mods or not isinstance(mods[0], HyperPatchNoPadding) or not isinstance(ref, HF.BankRef):
            return None
        conv = mods[0]
        if conv.kernel_size != (1, 1) or conv.stride != (1, 1) or conv.dilation != (1, 1) or conv.groups != 1 or ref.rows != conv.hyper_params:
            return None
        k, aff, act = 1, None, HF.ACT_NONE
        if k < len(mods) and isinstance(mods[k], nn.BatchNorm2d):
            bn = mods[k]
            if bn.training or (torch.is_grad_enabled() and bn.weight is not None and bn.weight.requires_grad):
                return None
            aff = seq._fold(k, bn)
            k += 1
        if k < len(mods) and _act_code(mods[k]) is not None:
            act = _act_code(mods[k])
            k += 1
        if k != len(mods):
            return None                                  # a Dropout or any other tail: the generic route
        sk = x[-l - 1]
        prev_c = couts[-1] if couts else 0
        if not (sk.is_cuda and sk.dtype == torch.float32 and sk.is_contiguous()) or 2 + sk.shape[1] + prev_c != conv.in_channels:
            return None
        skips.append(sk); bnk.append(ref.bank); couts.append(conv.out_channels); affines.append(aff); acts.append(act)
    if getattr(owner, '_k1_chain', None) is None:
        owner._k1_chain = HF.K1Chain()
    # the first inverted-residual level rides in the same launch when it is the block every reference configuration builds
    # (3 x 3 depthwise, stride 1, reflect halo, BatchNorm2d | identity, ReLU6, no residual) on 8 x 8-pixel patches
    ir = None
    if HF.K1_CHAIN_IR and len(seqs) > 3 and getattr(owner, 'chain_ir', HF.K1_CHAIN_IR_DEFAULT):
        mods = list(seqs[3])
        while len(mods) == 1 and isinstance(mods[0], MetaSequential):
            mods = list(mods[0])
        ref = refs[3]
        blk = mods[0] if len(mods) == 1 else None
        if isinstance(blk, HyperPatchInvertedResidual) and isinstance(ref, HF.BankRef) and ref.rows == blk.hyper_params \
                and not blk.use_res_connect and blk.kernel_size == (3, 3) and blk.stride == 1 and blk.padding_mode == 'reflect' \
                and isinstance(blk.act_layer, nn.ReLU6) \
                and all((isinstance(q, nn.BatchNorm2d) and not q.training and not (torch.is_grad_enabled() and q.weight.requires_grad))
                        or blk._is_identity(q) for q in (blk.bn1, blk.bn2, blk.bn3)):
            sk = x[-4]
            if sk.is_cuda and sk.dtype == torch.float32 and sk.is_contiguous() and 2 + sk.shape[1] + couts[2] == blk.in_nc:
                bns = [None if blk._is_identity(q) else blk._affine_of(k, q, sk.device) for k, q in enumerate((blk.bn1, blk.bn2, blk.bn3))]
                ir = dict(skip=sk, bank=ref.bank, hidden=blk.hidden_dim, c_out=blk.out_nc, bn=bns)
    if ir is not None:
        y = owner._k1_chain.run(skips, bnk, couts, affines, acts, ir=ir)
        if y is not None:
            return y, 4
    y = owner._k1_chain.run(skips, bnk, couts, affines, acts)
    return (y, 3) if y is not None else None


class MultiScaleDecoder(EpochOnModeSwitch, nn.Module):
    """Dynamic multi-scale decoder (hyperseg_v1_0.py:94-253).  ``forward(x, s)``: x = list of feature
    maps fine -> coarse including the input image, s = signal (B, Cs, H/32, W/32)."""

    def __init__(self, feat_channels, signal_channels, num_classes=3, kernel_sizes=3, level_layers=1,
                 level_channels=None, norm_layer=nn.BatchNorm2d, act_layer=nn.ReLU6(inplace=True), out_kernel_size=1,
                 expand_ratio=1, groups=1, weight_groups=1, with_out_fc=False, dropout=None, coords_res=None):
        super(MultiScaleDecoder, self).__init__()
        n = len(level_channels)
        kernel_sizes = per_level(kernel_sizes, n, 'kernel_sizes')
        level_layers = per_level(level_layers, n, 'level_layers')
        expand_ratio = per_level(expand_ratio, n, 'expand_ratio')
        if isinstance(groups, (list, tuple)):
            per_level(groups, n, 'groups')
        self.level_layers, self.levels = level_layers, n
        self.layer_params, self.coords_cache = [], {}
        self.weight_groups = weight_groups

        # modules: one MetaSequential per level, named level_<l> (the state-dict keys reference checkpoints carry)
        plan, carried = plan_levels(feat_channels, level_channels, kernel_sizes, level_layers, expand_ratio, groups,
                                    num_classes, with_out_fc)
        for lvl, layers in enumerate(plan):
            self.add_module(f'level_{lvl}', MetaSequential(*[self._make_layer(spec, norm_layer, act_layer) for spec in layers]))
        self.out_fc = None
        if with_out_fc:
            tail = [] if dropout is None else [nn.Dropout2d(dropout, True)]
            tail.append(HyperPatchConv2d(carried, num_classes, out_kernel_size, padding=out_kernel_size // 2))
            self.out_fc = MetaSequential(*tail)

        # what the context head sizes itself from: hyper-parameters per top-level consumer, and their running offsets
        self.param_groups = [getattr(self, f'level_{lvl}').hyper_params for lvl in range(n)]
        if self.out_fc is not None:
            self.param_groups.append(self.out_fc.hyper_params)
        offsets = np.cumsum([0] + self.param_groups)
        self._ranges = [int(v) for v in offsets[:n + 1]] + [int(offsets[-1])]

        register_coordinate_buffers(self, coords_res, n)      # checkpoint compatibility only

        # every signal-fed module gets a share of the signal channels proportional to its parameter count, in units of
        # the largest group count, and its own grouped 1x1 weight generator
        per_module = get_hyper_params(self)
        unit = max(weight_groups) if isinstance(weight_groups, (list, tuple)) else weight_groups
        shares = divide_feature(signal_channels, per_module, min_unit=unit)
        init_signal2weights(self, list(shares), weight_groups=weight_groups)
        self.hyper_params = sum(per_module)

    @staticmethod
    def _make_layer(spec, norm_layer, act_layer):
        if spec['k'] > 1:
            return HyperPatchInvertedResidual(spec['cin'], spec['cout'], spec['k'], expand_ratio=spec['expand'],
                                              norm_layer=norm_layer, act_layer=act_layer)
        return make_hyper_patch_conv2d_block(spec['cin'], spec['cout'], spec['k'], groups=spec['groups'])

    def cache_image_coordinates(self, h, w):
        return coordinate_grid(h, w)

    def get_image_coordinates(self, b, h, w, device):
        buf = getattr(self, f'coord{h}_{w}', None)
        grid = buf if buf is not None else coordinate_grid(h, w, device)
        return grid.expand(b, -1, -1, -1)

    def _hyper_modules(self):
        """Signal-fed modules per top-level child, in the order MetaSequential consumes weight-list entries."""
        if getattr(self, '_hyper_cache', None) is None:
            def collect(m):
                out = []
                for _, c in m.named_children():
                    out.extend([c] if isinstance(c, _HYPER_TYPES) else collect(c))
                return out
            found = [collect(getattr(self, f'level_{l}')) for l in range(self.levels)]
            found.append(collect(self.out_fc) if self.out_fc is not None else [])
            self._hyper_cache = found                          # published complete (replica threads may race to build it)
        return self._hyper_cache

    def _coschedule_plan(self, groups, layers):
        """Which k = 1 level's launch carries which later group's signal2weights layers: {level: [group indices]}, or None when the
        decoder does not start with a k = 1 level.  A group must be carried by a level BEFORE the one that consumes it; among the
        allowed carriers the least loaded takes it (cost ~ bank rows x (K + 32) per patch), the latest on a tie -- at HyperSeg-M:
        level 0 carries banks 1 + 2, level 1 bank 3, level 2 bank 4."""
        n_k1 = 0
        for l, g in enumerate(groups[:self.levels]):
            if len(g) == 1 and isinstance(g[0], HyperPatchNoPadding) and g[0].kernel_size == (1, 1):
                n_k1 += 1
            else:
                break
        if n_k1 == 0 or len(groups) < 2 or any(len(g) == 0 for g in groups[:self.levels]):
            return None
        starts, k0 = [], 0
        for g in groups:
            starts.append(k0)
            k0 += len(g)
        load = [0.0] * n_k1
        carry = {}
        for gi in range(len(groups) - 1, 0, -1):
            ls = layers[starts[gi]:starts[gi] + len(groups[gi])]
            if not ls:
                continue
            cost = sum(l['rows'] * (l['signal_channels'] / l['groups'] + 32.0) for l in ls)
            cands = range(0, min(gi, n_k1))
            best = min(cands, key=lambda j: (load[j], -j))
            load[best] += cost
            carry.setdefault(best, []).insert(0, gi)
        return carry

    def _run_k1_chain(self, x, banks):
        """Levels 0-2 (+ the first inverted-residual level) through functional.K1Chain: (output, levels done) or None."""
        seqs = [getattr(self, f'level_{l}') for l in range(min(self.levels, 4))]
        refs = [b[0] if len(b) == 1 else None for b in banks[:len(seqs)]]
        return run_decoder_chain(self, seqs, refs, x)

    def _train_banks(self, s):
        """Every level's bank for the training path in ONE launch (autograd.S2WBanksTrain: hs_s2w_train_fwd, three launches back) --
        or None when a level cannot take it (several signal-fed modules in a level, a signal2weights with a bias, K > 80, CPU)."""
        if not (HA.USE_HIP_S2W_TRAIN and s.is_cuda and s.dtype in (torch.float32, torch.bfloat16)):
            return None
        groups = self._hyper_modules()
        if any(len(g) != 1 for g in groups[:self.levels]) or (self.out_fc is not None and len(groups[-1]) != 1):
            return None
        flat = [g[0] for g in groups if g]
        b, _, fh, fw = s.shape
        if len(flat) > HF.S2W_TRAIN_MAX_LAYERS:                      # hs_s2w_train_*: S2W_MAX_LAYERS of include/hyperseg_hip.h
            return None
        meta, weights = [], []
        for m in flat:
            conv = m.signal2weights
            if conv is None or conv.bias is not None or m.signal_channels // conv.groups > 80 or conv.weight.dtype != torch.float32:
                return None
            # the per-module route hands a level MetaSequential's clamped slice s[:, :hyper_params] (meta_sequential.py:35, Appendix D-2)
            # and the inference route raises past it: the single launch must not read channels that slice would not contain
            if m.signal_index + m.signal_channels > min(int(m.hyper_params), s.shape[1]):
                return None
            # 32-bit element offsets inside hs_s2w_train_*: the level's bank and its signal2weights output
            if b * fh * fw * HF._round_up(int(m.hyper_params), 4) >= 2 ** 31 or b * conv.weight.shape[0] * fh * fw >= 2 ** 31:
                return None
            meta.append(dict(signal_index=m.signal_index, signal_channels=m.signal_channels, groups=conv.groups, rows=int(m.hyper_params)))
            weights.append(conv.weight.view(conv.weight.shape[0], -1))
        banks = HA.S2WBanksTrain.apply(meta, s, *weights)
        return [HF.TrainBank(bk, b, mt['rows'], (fh, fw)) for bk, mt in zip(banks, meta)]

    def _forward_autograd(self, x, s):
        """Training / gradient path: same modules, everything through autograd; the banks of all levels from one launch where the
        decoder's structure allows (_train_banks), per-module weight generation otherwise."""
        tb = self._train_banks(s) if isinstance(s, torch.Tensor) else None
        p = None
        for level in range(self.levels):
            p = getattr(self, f'level_{level}')(HF.StageInput(x[-level - 1], p, coords=True), tb[level] if tb is not None else s)
        if self.out_fc is not None:
            p = self.out_fc(p, tb[-1] if tb is not None else s)
        if p.shape[2:] != x[0].shape[2:]:
            p = HA.upsample_bilinear(p, x[0].shape[2:])
        return p

    def forward(self, x, s, masks=False):
        """``masks=True`` (inference only, not in the reference): uint8 argmax masks straight from the final upsample
        kernel instead of logits."""
        if self.training or HA.needs_grad(s, *x, *self.parameters()):
            assert not masks, 'masks=True is an inference-only shortcut'
            return self._forward_autograd(x, s)
        # every level's filter bank in ONE launch (the banks only depend on the signal)
        groups = self._hyper_modules()
        if any(len(g) > 1 for g in groups):
            # several signal-fed modules inside one level (level_layers > 1; no shipped config): the reference hands the
            # k-th of them the signal slice that starts at the hyper-parameter count of its predecessors
            # (meta_sequential.py:35), so their signal_index is relative to that slice -- take the per-module route,
            # which slices exactly like that, instead of the single launch over the whole signal
            assert not masks, 'masks=True needs the single-launch route'
            return self._forward_autograd(x, s)
        flat = [m for g in groups for m in g]
        for m in flat:
            # the reference hands each module s[:, 0:hyper_params] (MetaSequential's clamped slice, Appendix D-2)
            if m.signal_index + m.signal_channels > min(int(m.hyper_params), s.shape[1]):
                raise ValueError('signal slice of a decoder level exceeds what MetaSequential would hand to it')
        # The banks depend on the signal only.  The light k=1 levels' banks are produced on the current stream; the
        # heavy k=3 levels' banks (80 % of the signal2weights work) on a side stream, overlapping the k=1 levels.
        n_early = sum(len(g) for l, g in enumerate(groups)
                      if l < self.levels and all(not isinstance(m, HyperPatchInvertedResidual) for m in g)
                      and all(all(not isinstance(q, HyperPatchInvertedResidual) for q in groups[e]) for e in range(l)))
        layers = [m.s2w_layer(s.device) for m in flat]
        # coarse k = 1 levels generate their bank inside the consumer: no bank in HBM for them
        fh, fw = s.shape[-2:]
        in_consumer = {}
        for lvl, g in enumerate(groups[:self.levels]):
            hl, wl = x[-lvl - 1].shape[-2:]
            if len(g) == 1 and isinstance(g[0], HyperPatchNoPadding) and g[0].groups == 1 and \
                    hl % fh == 0 and wl % fw == 0 and (hl // fh) * (wl // fw) <= HF.BANK_IN_CONSUMER_MAX_PIXELS and \
                    g[0].signal_channels // g[0].signal2weights.groups <= 80:
                in_consumer[id(g[0])] = HF.SignalRef(s, g[0].s2w_layer(s.device))
        if in_consumer:
            keep = [i for i, m in enumerate(flat) if id(m) not in in_consumer]
            made = HF.signal2weights_multi(s, [layers[i] for i in keep]) if keep else []
            refs = [in_consumer.get(id(m)) for m in flat]
            for i, r in zip(keep, made):
                refs[i] = r
            banks, k = [], 0
            for g in groups:
                banks.append(refs[k:k + len(g)])
                k += len(g)
            p = None
            for level in range(self.levels):
                p = getattr(self, f'level_{level}')(HF.StageInput(x[-level - 1], p, coords=True), banks[level])
            if self.out_fc is not None:
                p = self.out_fc(p, banks[-1])
            if masks:
                return HF.upsample_argmax(p, x[0].shape[2:])
            if p.shape[2:] != x[0].shape[2:]:
                p = HF.upsample_bilinear(p, x[0].shape[2:], out=getattr(self, 'output_buffer', None))
            return p
        carry = self._coschedule_plan(groups, layers) if HF.COSCHEDULE_BANKS and s.is_cuda and not HF.PIPELINE_BANKS \
            and not HF.USE_SIDE_STREAM else None
        if carry is not None:
            # Level 0's bank as its own launch; every later level's bank rides in the launch of an EARLIER k = 1 level
            # (HF.CoScheduledBanks -> hs_patch_conv_s2w_fwd): the bank producer's blocks fill the CUs those latency-bound
            # launches leave idle, instead of standing in front of level 0 as one 15 us launch.
            per_group, k0 = [], 0
            for g in groups:
                per_group.append(layers[k0:k0 + len(g)])
                k0 += len(g)
            banks = [None] * len(groups)
            banks[0] = HF.signal2weights_multi(s, per_group[0])
            p = None
            for level in range(self.levels):
                stage = HF.StageInput(x[-level - 1], p, coords=True)
                riders = carry.get(level, [])
                if riders:
                    with HF.CoScheduledBanks(s, [l for gi in riders for l in per_group[gi]]) as co:
                        p = getattr(self, f'level_{level}')(stage, banks[level])
                    k1 = 0
                    for gi in riders:
                        banks[gi] = co.refs[k1:k1 + len(per_group[gi])]
                        k1 += len(per_group[gi])
                else:
                    p = getattr(self, f'level_{level}')(stage, banks[level])
            if self.out_fc is not None:
                p = self.out_fc(p, banks[-1])
            if masks:
                return HF.upsample_argmax(p, x[0].shape[2:])
            if p.shape[2:] != x[0].shape[2:]:
                p = HF.upsample_bilinear(p, x[0].shape[2:], out=getattr(self, 'output_buffer', None))
            return p
        side = join_level = None
        bank_events = {}
        if HF.PIPELINE_BANKS and s.is_cuda and len(groups) > 1 and len(groups[0]) > 0:
            # level 0's bank on this stream; each later level's bank = one launch on the side stream + one event: level l waits
            # for ITS bank only.  Bank memory is allocated here (this stream owns it); under capture the events become graph edges.
            main = torch.cuda.current_stream(s.device)
            fork = HF.SideStream.get(s.device)
            per_level, k = [], 0
            for g in groups:
                per_level.append(layers[k:k + len(g)])
                k += len(g)
            bufs = [torch.empty(HF.bank_floats(s, ls), device=s.device, dtype=torch.float32) if ls else None for ls in per_level]
            fork.wait_stream(main)
            refs = HF.signal2weights_multi(s, per_level[0], buf=bufs[0])
            with torch.cuda.stream(fork):
                for l in range(1, len(per_level)):
                    if per_level[l]:
                        refs = refs + HF.signal2weights_multi(s, per_level[l], buf=bufs[l])
                        ev = torch.cuda.Event()
                        ev.record(fork)
                        bank_events[l] = ev
        elif 0 < n_early < len(flat) and s.is_cuda and HF.USE_SIDE_STREAM:
            main = torch.cuda.current_stream(s.device)      # the MODEL's device, not the caller's current one
            side = HF.SideStream.get(s.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                late = HF.signal2weights_multi(s, layers[n_early:])
            refs = HF.signal2weights_multi(s, layers[:n_early]) + late
            cnt = 0
            for l, g in enumerate(groups):
                cnt += len(g)
                if cnt > n_early:
                    join_level = l
                    break
        else:
            refs = HF.signal2weights_multi(s, layers)
        banks, k = [], 0
        for g in groups:
            banks.append(refs[k:k + len(g)])
            k += len(g)
        p, first = None, 0
        if (getattr(self, 'chain_k1', False) or HF.K1_CHAIN) and side is None and not bank_events and s.is_cuda:
            # the three coarse k = 1 levels as ONE launch (hs_k1_chain_fwd); None: the shape / residency is not covered
            done = self._run_k1_chain(x, banks)
            p, first = done if done is not None else (None, 0)
        for level in range(first, self.levels):
            level_layers = getattr(self, f'level_{level}')
            if side is not None and level == join_level:
                torch.cuda.current_stream(s.device).wait_stream(side)
                side = None
            if level in bank_events:
                torch.cuda.current_stream(s.device).wait_event(bank_events.pop(level))
            # cat(coords, skip, bilinear(p)) is never built: the stage kernel's prologue generates it
            stage = HF.StageInput(x[-level - 1], p, coords=True)
            p = level_layers(stage, banks[level])
        if side is not None:
            torch.cuda.current_stream(s.device).wait_stream(side)
        for ev in bank_events.values():                     # (the out_fc's bank, or a level without hyper modules: join before leaving)
            torch.cuda.current_stream(s.device).wait_event(ev)
        if self.out_fc is not None:
            p = self.out_fc(p, banks[-1])
        if masks:
            return HF.upsample_argmax(p, x[0].shape[2:])
        if p.shape[2:] != x[0].shape[2:]:
            p = HF.upsample_bilinear(p, x[0].shape[2:], out=getattr(self, 'output_buffer', None))
        return p


class WeightMapper(nn.Module):
    """Context head (hyperseg_v1_0.py:379-448): 1x1 reduce, (levels-1) stride-2 2x2 convs down, global
    average at the bottom, 1x1 merges + nearest 2x up, concat -> signal.  Stock PyTorch-ROCm."""

    def __init__(self, in_channels, out_channels, levels=3, bias=False, min_unit=4, weight_groups=1):
        super().__init__()
        assert levels > 0, 'levels must be greater than zero'
        assert in_channels % 2 == 0, 'in_channels must be divisible by 2'
        if isinstance(weight_groups, (list, tuple)):
            assert len(weight_groups) == len(out_channels), f'groups ({len(weight_groups)}) must be of size {len(out_channels)}'
        self.in_channels, self.out_channels, self.levels = in_channels, out_channels, levels
        self.bias, self.weight_groups = bias, weight_groups
        half = in_channels // 2

        def block(cin, k, stride):
            return nn.Sequential(nn.Conv2d(cin, half, kernel_size=k, stride=stride, bias=bias),
                                 nn.BatchNorm2d(half), nn.ReLU(inplace=True))

        self.down_blocks = nn.ModuleList()
        self.up_blocks = nn.ModuleList()
        self.in_conv = block(in_channels, 1, 1)
        for _ in range(levels - 1):
            self.down_blocks.append(block(half, 2, 2))
            self.up_blocks.append(block(in_channels, 1, 1))
        self.upsample = nn.UpsamplingNearest2d(scale_factor=2)
        self._fused = None          # set by utils.inference.prepare_for_inference (single-frame inference route)

    def forward(self, x):
        if self._fused is not None and x.is_cuda and x.shape[0] == 1 and not self.training \
                and not (torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))) \
                and x.shape[2] % 2 ** self.levels == 0 and x.shape[3] % 2 ** self.levels == 0 \
                and (x.shape[2] * x.shape[3]) % 4 ** self.levels == 0:
            return self._fused(x.contiguous())
        feat = [self.in_conv(x)]
        for down in self.down_blocks:
            feat.append(down(feat[-1]))
        x = feat[-1]
        if x.shape[-2:] != (1, 1):
            x = F.adaptive_avg_pool2d(x, 1).expand_as(x)      # == avg pool + nearest resize back
        for level in range(self.levels - 2, -1, -1):
            x = self.upsample(self.up_blocks[level](torch.cat((feat.pop(), x), dim=1)))
        return torch.cat((feat.pop(), x), dim=1)


class HyperGen(HyperGenBase):
    """backbone -> context head -> dynamic decoder (hyperseg_v1_0.py:12-91); inference modes in HyperGenBase."""

    def __init__(self, backbone, weight_mapper, in_nc=3, num_classes=3, kernel_sizes=3, level_layers=1,
                 level_channels=None, expand_ratio=1, groups=1, weight_groups=1, inference_hflip=False,
                 inference_gather='mean', with_out_fc=False, decoder_groups=1, decoder_dropout=None, coords_res=None):
        super(HyperGen, self).__init__()
        self.inference_hflip, self.inference_gather = inference_hflip, inference_gather
        self.backbone = backbone()
        taps = self.backbone.feat_channels
        wg = list(weight_groups) if isinstance(weight_groups, (list, tuple)) else weight_groups   # the decoder pops from it
        self.decoder = MultiScaleDecoder([in_nc] + taps[:-1], taps[-1], num_classes, kernel_sizes, level_layers,
                                         level_channels, with_out_fc=with_out_fc, out_kernel_size=1,
                                         expand_ratio=expand_ratio, groups=decoder_groups, weight_groups=wg,
                                         dropout=decoder_dropout, coords_res=coords_res)
        self.weight_mapper = weight_mapper(taps[-1], self.decoder.param_groups)


def hyperseg_efficientnet(model_name, pretrained=False, out_feat_scale=0.25, levels=3, weights_path=None, **kwargs):
    """Config-file factory with the reference's signature (hyperseg_v1_0.py:813-827)."""
    from .backbones.efficientnet import efficientnet

    weight_mapper = partial(WeightMapper, levels=levels)
    backbone = partial(efficientnet, model_name, pretrained=pretrained, out_feat_scale=out_feat_scale, head=None,
                       return_features=True)
    model = HyperGen(backbone, weight_mapper, **kwargs)
    if weights_path is not None:
        checkpoint = torch.load(weights_path, map_location='cpu', weights_only=False)
        model.load_state_dict(checkpoint['state_dict'], strict=True)
    return model
