"""HyperSeg v1.0 "unify" on the MI355X decoder path -- drop-in for hyperseg/models/hyperseg_v1_0_unify.py
(HyperSeg-S Cityscapes 1536x768: configs/train/cityscapes_efficientnet_b1_hyperseg-s.py:10, 36-40).

Differences to v1_0, as in the reference: the ``signal2weights`` convolutions live in :class:`WeightLayer` modules
(``decoder.weight_blocks.{i}``), levels >= ``unify_level - 1`` share ONE weight layer whose output is channel-sliced
per level (hyperseg_v1_0_unify.py:172-178, 242-249), and the level modules receive weights, not the signal.
Here every weight layer of the decoder runs in one ``hs_signal2weights_multi_fwd`` launch and the shared bank is
consumed in place through row-range views (no slice copies).
"""
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import autograd as HA
from .. import functional as HF
import numpy as np

from ._common import EpochOnModeSwitch, HyperGenBase, coordinate_grid, per_level, plan_levels, register_coordinate_buffers
from .hyperseg_v1_0 import (HyperPatch, HyperPatchConv2d, HyperPatchInvertedResidual, HyperPatchNoPadding,  # noqa: F401
                            WeightMapper, _SignalToWeights, divide_feature, make_hyper_patch_conv2d_block,
                            next_multiply)
from .layers.meta_sequential import MetaSequential


class WeightLayer(EpochOnModeSwitch, nn.Module, _SignalToWeights):
    """signal -> weights of one level (or of all unified levels): hyperseg_v1_0_unify.py:287-309."""

    def __init__(self, target_params):
        super(WeightLayer, self).__init__()
        self.target_params = target_params
        self._init_s2w_state()

    @property
    def hyper_params(self):
        return self.target_params

    def init_signal2weights(self, signal_channels, signal_index=0, groups=1):
        self._make_signal2weights(signal_channels, signal_index, groups, next_multiply(self.target_params, groups))

    def s2w_layer(self, device):
        return self._s2w_layer(self.target_params)

    def forward(self, s):
        return self.apply_signal2weights(s)


def get_hyper_params(model):
    found = []
    for _, m in model.named_children():
        if isinstance(m, WeightLayer):
            found.append(m.target_params)
        else:
            found.extend(get_hyper_params(m))
    return found


def init_signal2weights(model, signal_features, signal_index=0, weight_groups=1):
    """Same traversal as v1_0 but over WeightLayer modules, which are siblings in one ModuleList: here the
    running signal offset DOES accumulate (0 / 576 / 704 / 768 for HyperSeg-S)."""
    for _, m in model.named_children():
        if isinstance(m, WeightLayer):
            nc = signal_features.pop(0)
            g = weight_groups.pop(0) if isinstance(weight_groups, list) else weight_groups
            m.init_signal2weights(nc, signal_index, g)
            signal_index += nc
        else:
            init_signal2weights(m, signal_features, signal_index, weight_groups)


class MultiScaleDecoder(EpochOnModeSwitch, nn.Module):
    """hyperseg_v1_0_unify.py:96-259."""

    def __init__(self, feat_channels, signal_channels, num_classes=3, kernel_sizes=3, level_layers=1,
                 level_channels=None, norm_layer=nn.BatchNorm2d, act_layer=nn.ReLU6(inplace=True), out_kernel_size=1,
                 expand_ratio=1, groups=1, weight_groups=1, with_out_fc=False, dropout=None,
                 coords_res=None, unify_level=None):
        super(MultiScaleDecoder, self).__init__()
        n = len(level_channels)
        kernel_sizes = per_level(kernel_sizes, n, 'kernel_sizes')
        level_layers = per_level(level_layers, n, 'level_layers')
        expand_ratio = per_level(expand_ratio, n, 'expand_ratio')
        self.level_layers, self.levels, self.unify_level = level_layers, n, unify_level
        self.layer_params, self.coords_cache = [], {}
        self.weight_groups = weight_groups

        plan, carried = plan_levels(feat_channels, level_channels, kernel_sizes, level_layers, expand_ratio, groups,
                                    num_classes, with_out_fc)
        self.level_blocks = nn.ModuleList(MetaSequential(*[self._make_layer(spec, norm_layer, act_layer) for spec in layers])
                                          for layers in plan)
        # weight generators: one per level below the unification point, then ONE for all remaining levels together,
        # whose output is cut into per-level channel ranges (self._ranges)
        first_shared = unify_level - 1
        shared = [b.hyper_params for b in self.level_blocks[first_shared:]]
        self.weight_blocks = nn.ModuleList([WeightLayer(b.hyper_params) for b in self.level_blocks[:first_shared]]
                                           + [WeightLayer(sum(shared))])
        self._ranges = [int(v) for v in np.cumsum([0] + shared)]

        self.out_fc = None
        if with_out_fc:
            tail = [] if dropout is None else [nn.Dropout2d(dropout, True)]
            tail.append(HyperPatchConv2d(carried, num_classes, out_kernel_size, padding=out_kernel_size // 2))
            self.out_fc = MetaSequential(*tail)

        register_coordinate_buffers(self, coords_res, self.levels)      # checkpoint compatibility only

        self.param_groups = get_hyper_params(self)
        min_unit = max(weight_groups) if isinstance(weight_groups, (list, tuple)) else weight_groups
        signal_features = divide_feature(signal_channels, self.param_groups, min_unit=min_unit)
        init_signal2weights(self, list(signal_features), weight_groups=weight_groups)
        self.hyper_params = sum(self.param_groups)

    @staticmethod
    def _make_layer(spec, norm_layer, act_layer):
        if spec['k'] > 1:
            return HyperPatchInvertedResidual(spec['cin'], spec['cout'], spec['k'], expand_ratio=spec['expand'],
                                              norm_layer=norm_layer, act_layer=act_layer)
        return make_hyper_patch_conv2d_block(spec['cin'], spec['cout'], spec['k'], groups=spec['groups'])

    def cache_image_coordinates(self, h, w):
        return coordinate_grid(h, w)

    def _forward_autograd(self, x, s):
        """Training / gradient path: weight layers as stock grouped 1x1 convs, levels through hyperseg_amd.autograd."""
        ul = self.unify_level
        p, w = None, None
        for level in range(self.levels):
            stage = HF.StageInput(x[-level - 1], p, coords=True)
            wb = self.weight_blocks[min(level, ul - 1)]
            if level <= ul - 1:
                w = wb._weights_train(s)
            if level < ul - 1:
                p = self.level_blocks[level](stage, w)
            else:
                i = level - ul + 1
                p = self.level_blocks[level](stage, w[:, self._ranges[i]:self._ranges[i + 1]])
        if p.shape[2:] != x[0].shape[2:]:
            p = HA.upsample_bilinear(p, x[0].shape[2:])
        return p

    def forward(self, x, s, masks=False):
        if self.out_fc is not None:
            raise NotImplementedError('with_out_fc=True: the reference itself feeds the raw signal to out_fc here '
                                      '(hyperseg_v1_0_unify.py:252-253); no config uses it')
        if self.training or HA.needs_grad(s, *x, *self.parameters()):
            assert not masks, 'masks=True is an inference-only shortcut'
            return self._forward_autograd(x, s)
        wl = list(self.weight_blocks)
        ul = self.unify_level
        layers = [m.s2w_layer(s.device) for m in wl]
        # coarse k = 1 levels with a weight layer of their own generate their bank inside the consumer; every other
        # weight layer is produced by one launch
        fh, fw = s.shape[-2:]
        refs = [None] * len(wl)
        # the chained launch (levels 0-2) reads materialised banks: with it on, those levels' banks come from the one signal2weights
        # launch instead of being generated inside their consumers (the SignalRefs made the hook below dead)
        chain = (getattr(self, 'chain_k1', False) or HF.K1_CHAIN) and s.is_cuda and ul - 1 >= 3 and \
            not (getattr(self, '_k1_chain', None) is not None and getattr(self._k1_chain, 'refuses_everything', False))
        for lvl in range(min(ul - 1, self.levels)):
            if chain and lvl < 3:
                continue
            mods = [m for m in self.level_blocks[lvl][0].children()] if len(self.level_blocks[lvl]) == 1 and \
                isinstance(self.level_blocks[lvl][0], MetaSequential) else []
            hl, wdt = x[-lvl - 1].shape[-2:]
            if mods and isinstance(mods[0], HyperPatchNoPadding) and mods[0].groups == 1 and \
                    hl % fh == 0 and wdt % fw == 0 and (hl // fh) * (wdt // fw) <= HF.BANK_IN_CONSUMER_MAX_PIXELS and \
                    layers[lvl]['signal_channels'] // layers[lvl]['groups'] <= 80:
                refs[lvl] = HF.SignalRef(s, layers[lvl])
        keep = [i for i, r in enumerate(refs) if r is None]
        for i, r in zip(keep, HF.signal2weights_multi(s, [layers[i] for i in keep])):
            refs[i] = r
        p, first = None, 0
        if chain:
            # levels 0-2 have weight layers of their own: one launch for the three of them (hs_k1_chain_fwd); None: shape / residency not covered
            from .hyperseg_v1_0 import run_decoder_chain
            done = run_decoder_chain(self, [self.level_blocks[l] for l in range(3)], refs[:3], x)
            p, first = done if done is not None else (None, 0)
        for level in range(first, self.levels):
            stage = HF.StageInput(x[-level - 1], p, coords=True)
            if level < ul - 1:
                w = refs[level]
            else:
                i = level - ul + 1
                shared = refs[ul - 1]
                r0, r1 = self._ranges[i], self._ranges[i + 1]
                # rows [r0, r1) of the shared bank, consumed in place (reference: w[:, r0:r1] + .contiguous())
                w = HF.BankRef(shared.bank[:, r0:r1], shared.shape[0], r1 - r0, shared.grid)
            p = self.level_blocks[level](stage, [w])
        if masks:
            return HF.upsample_argmax(p, x[0].shape[2:])
        if p.shape[2:] != x[0].shape[2:]:
            p = HF.upsample_bilinear(p, x[0].shape[2:], out=getattr(self, 'output_buffer', None))
        return p


class HyperGen(HyperGenBase):
    """hyperseg_v1_0_unify.py:12-93; inference modes in HyperGenBase."""

    def __init__(self, backbone, weight_mapper, in_nc=3, num_classes=3, kernel_sizes=3, level_layers=1,
                 level_channels=None, expand_ratio=1, groups=1, weight_groups=1, inference_hflip=False,
                 inference_gather='mean', with_out_fc=False, decoder_groups=1, decoder_dropout=None, coords_res=None,
                 unify_level=None):
        super(HyperGen, self).__init__()
        self.inference_hflip, self.inference_gather = inference_hflip, inference_gather
        self.backbone = backbone()
        taps = self.backbone.feat_channels
        wg = list(weight_groups) if isinstance(weight_groups, (list, tuple)) else weight_groups
        self.decoder = MultiScaleDecoder([in_nc] + taps[:-1], taps[-1], num_classes, kernel_sizes, level_layers,
                                         level_channels, with_out_fc=with_out_fc, out_kernel_size=1,
                                         expand_ratio=expand_ratio, groups=decoder_groups, weight_groups=wg,
                                         dropout=decoder_dropout, coords_res=coords_res, unify_level=unify_level)
        self.weight_mapper = weight_mapper(taps[-1], self.decoder.param_groups)


def hyperseg_efficientnet(model_name, pretrained=False, out_feat_scale=0.25, levels=3, weights_path=None, **kwargs):
    """Config-file factory (hyperseg_v1_0_unify.py:654-668)."""
    from .backbones.efficientnet import efficientnet

    weight_mapper = partial(WeightMapper, levels=levels)
    backbone = partial(efficientnet, model_name, pretrained=pretrained, out_feat_scale=out_feat_scale, head=None,
                       return_features=True)
    model = HyperGen(backbone, weight_mapper, **kwargs)
    if weights_path is not None:
        checkpoint = torch.load(weights_path, map_location='cpu', weights_only=False)
        model.load_state_dict(checkpoint['state_dict'], strict=True)
    return model
