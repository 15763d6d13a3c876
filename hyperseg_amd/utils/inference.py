"""Inference-time preparation of the encoder and the context head of a HyperGen model (the rows
next to the decoder hot path).  Opt-in: ``prepare_for_inference(model, fused_depthwise=True)`` after ``model.eval()``
and after loading a checkpoint; modules, parameters and state-dict keys are left untouched -- the fused routes read the
model's own parameters and keep their folded BatchNorm affines in non-persistent buffers.

``fused_depthwise=True`` (historical name) switches on, for CUDA tensors in eval mode:
  * ``FusedStem``        conv_stem + BN + swish                           -> hs_stem_conv_fwd
  * ``FusedMBConv``      every MBConv block in 4-5 launches               -> hs_mbconv_expand_dw_fwd | library GEMM +
                         hs_depthwise_conv_fwd, hs_se_gate_fwd, library GEMM (gate / BN folded into its weights,
                         BN shift deferred to the consumers), see the class docstring
  * ``FusedPointwise``   feature reducers and the head conv              -> hs_pointwise_conv_fwd | GEMM + hs_affine_act_fwd
  * ``FusedContextHead`` the v1_0 WeightMapper without its concatenations -> library GEMMs + hs_affine_act_fwd
``split_gemm=True`` (with ``fused_depthwise``; what bench.py runs since round 3: whole-model parity on the GPU in
tests/test_split_gemm.py and test_benched_configuration_replay_matches_stock, 1091 -> 1179 frames/s at HyperSeg-M) sends the
MBConv blocks' expand / project 1x1 convolutions and the GEMM-routed reducers through ``hs_gemm_split_fwd`` (f16 matrix cores,
split operands, SE gate applied inside) instead of the library f32 GEMM.
``fold_bn=True`` additionally folds the remaining Conv -> BatchNorm pairs of the stock modules into the convolutions
(that one DOES change the state dict: BN entries turn into identities) and ``channels_last`` switches the stock encoder's
memory format; neither is used by bench.py.
DESIGN.md section 6c has the measurements behind every routing threshold below."""
import contextlib
import os

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


# output pixels from which the fused expand + depthwise kernel is used (tools/bench_mbconv.py measures both routes)
# round 6: the split GEMM covers 1280 < K <= 2560 on its two-chunk form (tests/test_split_gemm.py), but HyperSeg-M's one such layer -- the last
# project conv, K = 1920 -- measured 1.8 us SLOWER on it than on the library GEMM (profiles/round6_split_gemm_k1920_negative_w14.txt)
SPLIT_GEMM_MAX_K = int(os.environ.get('HS_SPLIT_GEMM_MAX_K', '1280'))
STEM_DW = os.environ.get('HS_STEM_DW', '1') != '0'       # stem + block 0's depthwise half as one launch (hs_stem_dw_fwd); 0: two launches
FUSE_EXPAND_MIN_PIXELS = int(os.environ.get('HS_FUSE_EXPAND_MIN_PIXELS', '2048'))
FUSE_EXPAND_MAX_CIN = int(os.environ.get('HS_FUSE_EXPAND_MAX_CIN', '40'))      # wider inputs: 1 wave / SIMD, slower than GEMM + dw


# batch-1 project convs need no epilogue once gate / BN scale are folded into the weights: the bare library GEMM beats
# hs_pointwise_conv_fwd for every K but the smallest (tools/bench_mbconv.py, second table)
LEAN_MFMA_MAX_CIN = int(os.environ.get('HS_LEAN_MFMA_MAX_CIN', '16'))
PW_MFMA_MAX_CIN = int(os.environ.get('HS_PW_MFMA_MAX_CIN', '96'))
PW_MFMA_MIN_PIXELS = int(os.environ.get('HS_PW_MFMA_MIN_PIXELS', '8192'))
GEMM_LT_MIN_PIXELS = int(os.environ.get('HS_GEMM_LT_MIN_PIXELS', '16384'))       # see gemm_library


@contextlib.contextmanager
def gemm_library(pixels, batch=1):
    """Which BLAS backs the bare fp32 GEMMs of the 1x1 convolutions (ROCm 7.2, measured per layer under graph replay,
    profiles/round1_frame_sequence*.txt): hipBLASLt for the few-channel / many-pixel project convs of the first stages
    (24 x 32768 x 144: 9.8 us vs rocBLAS 21.3), rocBLAS for everything at <= 64x128 pixels (672 x 2048 x 112: 9.1 us vs
    hipBLASLt 19.5).  Strided-BATCHED GEMMs (batch > 1): hipBLASLt throughout (HyperSeg-L bs 32: 10.50 ms per batch with
    hipBLASLt everywhere, 10.77 with the single-frame rule, 11.68 with rocBLAS everywhere; gpurun r3h).
    torch's names: 'cublaslt' = hipBLASLt, 'cublas' = rocBLAS.  The choice is made at call (= capture) time."""
    want = os.environ.get('HS_BLAS') or ('cublaslt' if (pixels >= GEMM_LT_MIN_PIXELS or batch > 1) else 'cublas')
    prev = torch.backends.cuda.preferred_blas_library()
    torch.backends.cuda.preferred_blas_library(want)
    try:
        yield
    finally:
        torch.backends.cuda.preferred_blas_library(prev)


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
        self.split_gemm = False          # prepare_for_inference(split_gemm=True): hs_gemm_split_fwd instead of the library GEMM
        self._split = {}                 # (with BN scale, device) -> functional.SplitWeights | None (Cin not covered)

    @property
    def conv(self):
        return self._conv[0]

    def split_weights(self, with_scale, device):
        """The conv weight prepared for hs_gemm_split_fwd (built once per FusedPointwise, i.e. again after every
        load_state_dict -- _install_fused recreates these modules); ``with_scale``: BN scale folded into the rows."""
        from .. import functional as HF
        srcs = (self.conv.weight, self.scale) if with_scale else (self.conv.weight,)
        key, ver = (bool(with_scale), device), HF._key(*srcs)       # in-place updates of the weight / BN scale are seen
        hit = self._split.get(key)
        if hit is None or hit[0] != ver:
            hit = (ver, HF.gemm_split_weights(self.conv.weight, self.scale if with_scale else None, max_k=SPLIT_GEMM_MAX_K), HF.producer_stream(self.conv.weight.device))
            HF.publish_ready(self.conv.weight.device)            # another stream / replica thread may pick the entry up
            self._split[key] = hit
        elif hit[1] is not None:
            HF.adopt((hit[1].frag, hit[1].inv), self.conv.weight.device, hit[2])
        return hit[1]

    @torch.no_grad()
    def absorb_input_offset(self, offset):
        """The input arrives as x - offset (a per-channel constant its producer did not add, see FusedMBConv): a 1x1
        conv is linear, so the missing term is a constant per output channel -- fold it into the BN shift."""
        w = self.conv.weight.detach().flatten(1).to(offset.device)
        self.shift.add_(self.scale * (w @ offset))

    def uses_mfma(self, x):
        """Small-K, many-pixel layers run as one fused MFMA GEMM; large-K layers keep the library GEMM
        (tools/bench_mbconv.py times both routes per layer)."""
        return self.mfma_covers(x.shape[1], x.shape[2] * x.shape[3])

    @staticmethod
    def mfma_covers(channels, pixels):
        return channels <= PW_MFMA_MAX_CIN and pixels >= PW_MFMA_MIN_PIXELS

    def raw(self, x):
        """The bare GEMM: W (Cout, Cin) @ x (Cin, HW) per frame (a strided-batched GEMM with a shared A for a batch); BN +
        activation are left to the consumer."""
        b, cin, h, w = x.shape
        if self.split_gemm:
            sw = self.split_weights(False, x.device)
            if sw is not None:
                from .. import functional as HF
                return HF.gemm_split(sw, x.contiguous())
        with gemm_library(h * w, b):
            if b == 1:
                return torch.mm(self.conv.weight.view(-1, cin), x.view(cin, h * w)).view(1, -1, h, w)
            # bmm with the weight expanded over the batch (stride 0): matmul(2-D, 3-D) would go through transposed copies
            return torch.bmm(self.conv.weight.view(1, -1, cin).expand(b, -1, -1), x.view(b, cin, h * w)).view(b, -1, h, w)

    def forward(self, x, gate=None, residual=None):
        """``gate`` (B, Cin): SE gate applied to the input.  Non-MFMA shapes: stock GEMM followed by ONE fused
        BatchNorm + activation + skip-add launch."""
        import torch.nn.functional as F
        from .. import functional as HF
        conv = self.conv
        x = x.contiguous()
        b, cin, h, w = x.shape
        if self.uses_mfma(x):
            return HF.pointwise_conv(x, conv.weight, gate, self.scale, self.shift, self.act, residual)
        if (h * w) % 4 != 0:
            raise NotImplementedError('feature maps with H*W % 4 != 0')
        sw = self.split_weights(True, x.device) if self.split_gemm else None
        if sw is not None:       # BN scale folded into the split weights; shift, activation and skip add in the GEMM's tail
            return HF.gemm_split(sw, x, gate=gate, shift=self.shift, act=self.act, residual=residual)
        if gate is None:
            # batches: one strided-batched GEMM (MIOpen would go NCHW -> NHWC -> implicit GEMM -> NCHW)
            y = F.conv2d(x, conv.weight) if b == 1 else self.raw(x)
        else:
            y = F.conv2d(x * gate[:, :, None, None], conv.weight)
        return HF.affine_act_(y, self.scale, self.shift, self.act, residual)


class FusedStem(nn.Module):
    """conv_stem (3x3 stride 2, TF-"SAME") + BN + swish as one ``hs_stem_conv_fwd`` launch (stock: pad fill + pad copy +
    MIOpen conv + BatchNorm + SiLU)."""

    def __init__(self, conv, bn):
        super().__init__()
        assert conv.kernel_size == (3, 3) and conv.stride == (2, 2) and conv.in_channels == 3 and conv.bias is None
        scale, shift = _bn_affine(bn)
        self.register_buffer('scale', scale, persistent=False)
        self.register_buffer('shift', shift, persistent=False)
        self._conv = [conv]
        if conv._pad is not None:
            self.pad_l, self.pad_t = conv._pad[0], conv._pad[2]
            self.pad_w, self.pad_h = conv._pad[0] + conv._pad[1], conv._pad[2] + conv._pad[3]
        else:
            self.pad_t, self.pad_l = conv.padding
            self.pad_h, self.pad_w = 2 * conv.padding[0], 2 * conv.padding[1]
        # the weight as the GEMM operand of hs_stem_dw_fwd: (Cout, 27) + one zero column (rebuilt with the module by _install_fused)
        with torch.no_grad():
            self.register_buffer('w28', nn.functional.pad(conv.weight.detach().flatten(1), (0, 1)).contiguous(), persistent=False)

    def out_size(self, x):
        h, w = x.shape[2:]
        return (h + self.pad_h - 3) // 2 + 1, (w + self.pad_w - 3) // 2 + 1

    def forward(self, x):
        from .. import functional as HF
        x = x.contiguous()
        return HF.stem_conv_bn_swish(x, self._conv[0].weight, self.pad_t, self.pad_l, self.out_size(x), self.scale, self.shift)


class FusedMBConv(nn.Module):
    """A whole MBConv block in 5 launches (stock: 15; MIOpen has no tuned fp32 depthwise solver on ROCm 7.2 -- its
    Winograd-per-group / naive kernels cost half of the frame).  Filters stay the block's own parameters.

    Early blocks (few channels, many pixels; any batch):
        [1x1 expand + BN + swish: MFMA] -> [depthwise + BN + swish + SE pooling] -> [SE squeeze] -> [SE excite] ->
        [gate * 1x1 project + BN + skip: MFMA]
    Late blocks (many channels, few pixels; batch 1): the 1x1 convs are plain library GEMMs with NOTHING after them --
        [expand GEMM, raw] -> [depthwise: BN0 + swish applied to the taps on load, + BN1 + swish + SE pooling] ->
        [SE squeeze] -> [SE excite + gate and BN2 scale folded into the project weights] ->
        [project GEMM, accumulating onto the skip tensor in place (beta = 1)]
      The BN2 shift is a per-channel constant; when every consumer of the block's output is a 1x1 conv (the next block's
      expand, the feature reducer, the head) it is not added at all: the block stores ``y - out_offset`` and the consumers
      fold ``W @ out_offset`` into their own BN shift (``absorb_input_offset``), exactly, offline.
    """

    def __init__(self, blk, in_offset=None, defer_shift=False):
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
        self._red_w = None                       # (Csq, C) view of the SE reduce weight
        # the gate by the pooling launch's last workgroups (round 5) -- only where the project convolution takes the gate as a vector
        # (our split GEMM / MFMA kernels); the library-GEMM route folds it into the weights and keeps hs_se_gate_fwd
        self.se_tail = True
        # block 0 only (round 6): the encoder's FusedStem, when stem + this block's depthwise half run as ONE launch (hs_stem_dw_fwd);
        # the block is then handed the IMAGE.  A list: the stem module stays registered once, under the backbone
        self._stem = None
        # constant-offset bookkeeping (see the class docstring)
        if in_offset is not None:
            assert self.expand is not None, 'a depthwise conv cannot consume an offset tensor (zero padding)'
            self.expand.absorb_input_offset(in_offset)
        with torch.no_grad():
            carried = self.project.shift.clone()
            if self.skip and in_offset is not None:
                carried += in_offset.to(carried.device)          # the skip tensor is short of in_offset as well
            self.defer_shift = bool(defer_shift)
            self.out_offset = carried.clone() if defer_shift else None
            self.project.shift.copy_(torch.zeros_like(carried) if defer_shift else carried)

    def fuses_expand(self, x, ho, wo):
        """[expand + BN + swish + depthwise + BN + swish + pool] as ONE launch (hs_mbconv_expand_dw_fwd): wherever the
        map is large enough to fill the chip with (tile x channel-chunk) workgroups and Cin fits the register-resident
        B fragments.  Small late maps keep the library GEMM + depthwise kernel pair."""
        return self.expand is not None and x.shape[1] <= FUSE_EXPAND_MAX_CIN and ho * wo >= FUSE_EXPAND_MIN_PIXELS

    def forward(self, inputs, blk):
        from .. import functional as HF
        pre = None
        if self._stem is not None and inputs.shape[1] == 3:      # inputs is the image: stem + depthwise half in one launch, or the stem's own launch first
            stem = self._stem[0]
            img = inputs.contiguous()
            pre = HF.stem_dw(img, stem.w28, stem.scale, stem.shift, stem.pad_t, stem.pad_l, stem.out_size(img),
                             blk._depthwise_conv.weight, self.pad_t, self.pad_l, self.scale, self.shift, pool=True) if STEM_DW else None
            inputs = pre[0] if pre is not None else stem(img)         # (pre: only the shape is read below -- skip is off for this block)
        x = inputs.contiguous()
        b, _, h, w = x.shape
        ho = (h + self.pad_h - self.k) // self.stride + 1
        wo = (w + self.pad_w - self.k) // self.stride + 1
        lean = True                               # library GEMMs with nothing around them (batched for b > 1)
        red, exp = blk._se_reduce, blk._se_expand
        if self._exp_t is None or self._exp_t.device != x.device:
            self._exp_t = exp.weight.detach().flatten(1).t().contiguous()
        if self._red_w is None or self._red_w.device != x.device:
            self._red_w = red.weight.detach().flatten(1)             # a view: (Csq, C)
        # round 5: the pooling launch finishes the squeeze-excite gate in its last workgroups (csrc/hs_se_tail.h) -- `gated` says
        # whether `pooled` is already the gate (B, C) or still the partial sums for HF.se_gate
        proj = self.project
        cmid = blk._depthwise_conv.weight.shape[0]
        lean_gemm = lean and (cmid > LEAN_MFMA_MAX_CIN if self.defer_shift else not proj.mfma_covers(cmid, ho * wo))
        folds = lean_gemm and not (proj.split_gemm and proj.split_weights(True, x.device) is not None)      # gate folded into the weights
        se = (self._red_w, red.bias, self._exp_t, exp.bias) if self.se_tail and not folds else None
        if pre is not None:
            out = (pre[0], pre[1], False) if se is not None else pre
        elif self.fuses_expand(x, ho, wo):
            out = HF.mbconv_expand_dw(x, self.expand.conv.weight, self.expand.scale, self.expand.shift,
                                      blk._depthwise_conv.weight, self.stride, self.pad_t, self.pad_l, (ho, wo),
                                      self.scale, self.shift, pool=True, se=se)
        else:
            in_scale = in_shift = None
            if self.expand is not None:
                if lean and not self.expand.uses_mfma(x):
                    in_scale, in_shift = self.expand.scale, self.expand.shift
                    x = self.expand.raw(x)
                else:
                    x = self.expand(x)
            out = HF.depthwise_conv_bn_act(x, blk._depthwise_conv.weight, self.stride, self.pad_t, self.pad_l,
                                           (ho, wo), self.scale, self.shift, act=3, pool=True,
                                           in_scale=in_scale, in_shift=in_shift, se=se)
        y, pooled, gated = out if se is not None else (out[0], out[1], False)

        def se_gate(**fold):
            if gated and not fold:
                return pooled
            return HF.se_gate(pooled, b, ho * wo, red.weight, red.bias, self._exp_t, exp.bias, **fold)
        skip = inputs.contiguous() if self.skip else None
        # deferred shift -> nothing follows the GEMM: the bare library GEMM wins for all but the smallest K; otherwise the
        # MFMA kernel with its fused epilogue wins wherever it applies (few channels, many pixels)
        if lean_gemm:
            sw = proj.split_weights(True, y.device) if proj.split_gemm else None
            if sw is not None:
                # our own GEMM (f16 matrix cores, split operands): the BN2 scale is folded into the static split weights, the
                # gate multiplies the rows of y on load -- no per-frame copy of the project weights is written
                gate = se_gate()
                if self.defer_shift:
                    if skip is None:
                        return HF.gemm_split(sw, y, gate=gate)
                    return HF.gemm_split(sw, y, gate=gate, residual=skip, out=skip)     # in place: no other consumer
                return HF.gemm_split(sw, y, gate=gate, shift=proj.shift, residual=skip)
            # gate (and BN2 scale) folded into the project weights by the SE kernel: ~1e5 weights instead of a pass over y
            wp = se_gate(w_proj=proj.conv.weight, out_scale=proj.scale)
            cmid = y.shape[1]
            with gemm_library(ho * wo, b):
                if b == 1:
                    w2d, y2d = wp.view(-1, cmid), y.view(cmid, ho * wo)
                    if self.defer_shift:
                        if skip is None:
                            return torch.mm(w2d, y2d).view(1, -1, ho, wo)
                        skip.view(-1, ho * wo).addmm_(w2d, y2d)      # in place: the block input has no other consumer
                        return skip
                    out = torch.mm(w2d, y2d).view(1, -1, ho, wo)
                else:                             # per-frame gated weights: one strided-batched GEMM
                    w3d, y3d = wp.view(b, -1, cmid), y.view(b, cmid, ho * wo)
                    if self.defer_shift:
                        if skip is None:
                            return torch.bmm(w3d, y3d).view(b, -1, ho, wo)
                        skip.view(b, -1, ho * wo).baddbmm_(w3d, y3d)
                        return skip
                    out = torch.bmm(w3d, y3d).view(b, -1, ho, wo)
            return HF.affine_act_(out, None, proj.shift, 0, skip)
        return proj(y, gate=se_gate(), residual=skip)


CTX_DOWN_SPLIT = os.environ.get('HS_CTX_DOWN_SPLIT', '1') != '0'      # A/B switch (tools/gpu_ab_env.sh): 0 = F.conv2d + affine for the head's down blocks


class FusedContextHead(nn.Module):
    """The v1_0 context head (WeightMapper, hyperseg_v1_0.py:379-448) for one frame with fewer launches -- same library
    GEMMs, but every Conv -> BatchNorm -> ReLU is GEMM + ONE affine/ReLU launch, and the concatenations are never built:

      * ``cat(feat, pooled.expand_as(feat))`` in front of a 1x1 conv: the pooled half is constant over the pixels, so its
        product with the right half of the weights is a per-channel constant -- one mat-vec folded into the BN shift;
      * the final ``cat(feat0, upsample(u))`` is written in place: the first GEMM's output IS the left half of the
        signal buffer, the nearest-2x upsample is a broadcast copy into the right half.

    Weights are read through the head's own parameters; folded BN affines are non-persistent buffers."""

    def __init__(self, wm):
        super().__init__()
        self._wm = [wm]
        blocks = [wm.in_conv] + list(wm.down_blocks) + list(wm.up_blocks)
        for i, blk in enumerate(blocks):
            assert isinstance(blk[0], nn.Conv2d) and isinstance(blk[1], nn.BatchNorm2d) and isinstance(blk[2], nn.ReLU)
            assert blk[0].bias is None
            scale, shift = _bn_affine(blk[1])
            self.register_buffer(f'scale{i}', scale, persistent=False)
            self.register_buffer(f'shift{i}', shift, persistent=False)
        self.n_down = n = len(wm.down_blocks)
        # deepest merge: BN scale folded into the half of the weights that multiplies the (pixel-constant) pooled vector
        with torch.no_grad():
            up = wm.up_blocks[n - 1]
            half = up[0].out_channels
            wb = up[0].weight.detach().flatten(1)[:, half:]
            self.register_buffer('wb_scaled', getattr(self, f'scale{2 * n}')[:, None] * wb, persistent=False)

        self.split_gemm = False          # prepare_for_inference(split_gemm=True): the 1x1 convolutions through hs_gemm_split_fwd
        self._split = {}                 # name -> (key of the sources, SplitWeights | None)

    def _affine(self, i):
        return getattr(self, f'scale{i}'), getattr(self, f'shift{i}')

    def _split_weights(self, name, weight, scale, max_k=1280):
        """``weight`` (Cout, K) with the BN ``scale`` folded into its rows, prepared for hs_gemm_split_fwd (None when K is outside
        what the kernel covers); rebuilt when the conv weight or the BN scale change in place (functional._key)."""
        from .. import functional as HF
        key = HF._key(weight, scale)
        hit = self._split.get(name)
        if hit is None or hit[0] != key:
            hit = (key, HF.gemm_split_weights(weight, scale, max_k=max_k), HF.producer_stream(weight.device))
            HF.publish_ready(weight.device)
            self._split[name] = hit
        elif hit[1] is not None:
            HF.adopt((hit[1].frag, hit[1].inv), weight.device, hit[2])
        return hit[1]

    def forward(self, x):
        with gemm_library(x.shape[2] * x.shape[3]):
            return self._forward(x)

    def _forward(self, x):
        from .. import functional as HF
        import torch.nn.functional as F
        wm = self._wm[0]
        _, cin, h, w = x.shape
        half = cin // 2
        n = self.n_down
        signal = torch.empty(1, cin, h, w, device=x.device, dtype=torch.float32)
        # feat0 = relu(bn(in_conv(x))), produced directly as the left half of the signal
        left = signal[:, :half]
        sw_in = self._split_weights('in', wm.in_conv[0].weight.view(half, cin), self.scale0) if self.split_gemm and (h * w) % 4 == 0 else None
        if sw_in is not None:            # Conv -> BN -> ReLU in ONE launch: BN scale in the split weights, shift + ReLU in the GEMM's tail
            HF.gemm_split(sw_in, x.contiguous(), shift=self.shift0, act=HF.ACT_RELU, out=left)
        else:
            torch.mm(wm.in_conv[0].weight.view(half, cin), x.view(cin, h * w), out=left.view(half, h * w))
            HF.affine_act_(left, *self._affine(0), HF.ACT_RELU)
        feat = [left]
        pool = None
        for i, down in enumerate(wm.down_blocks):
            src = feat[-1]
            sw_dn = None
            if self.split_gemm and CTX_DOWN_SPLIT and src.shape[2] % 2 == 0 and src.shape[3] % 4 == 0 and src.is_contiguous() and src.data_ptr() % 16 == 0:
                sw_dn = self._split_weights(f'down{i}', down[0].weight, getattr(self, f'scale{1 + i}'), max_k=2560)
            if sw_dn is not None:        # the 2x2 / stride-2 conv -> BN -> ReLU in ONE launch, the window read on load (no im2col copy)
                ho, wo = src.shape[2] // 2, src.shape[3] // 2
                if i == n - 1 and (ho, wo) != (1, 1):     # the bottom: its global average comes out of the same launch as block sums
                    pool = torch.empty(1, half, -(-ho * wo // 16), device=x.device, dtype=torch.float32)
                feat.append(HF.gemm_split_conv2x2(sw_dn, src, shift=getattr(self, f'shift{1 + i}'), act=HF.ACT_RELU, pool_partial=pool))
            else:
                t = F.conv2d(src, down[0].weight, stride=2)
                feat.append(HF.affine_act_(t, *self._affine(1 + i), HF.ACT_RELU))
        t = feat[-1]
        pooled_const = t.shape[-2:] != (1, 1)            # the bottom is replaced by its global average
        for level in range(n - 1, -1, -1):
            up = wm.up_blocks[level]
            scale, shift = self._affine(1 + n + level)
            wgt = up[0].weight.view(half, cin)
            skip = feat.pop()                             # feat[level + 1]
            sh, sw = skip.shape[-2:]
            if pooled_const:
                # right operand is constant over the pixels: W_b @ mean -> per-channel constant in the shift
                if pool is not None:
                    shift = HF.pooled_shift(pool, t.shape[2] * t.shape[3], self.wb_scaled, shift)      # mean + mat-vec, one launch
                else:
                    shift = torch.addmv(shift, self.wb_scaled, t.mean((2, 3)).view(half))
                sw_up = self._split_weights(f'up{level}', wgt[:, :half], scale) if self.split_gemm and (sh * sw) % 4 == 0 else None
                if sw_up is not None and level == 0:
                    # ... and the nearest 2x upsample into the right half of the signal is the GEMM's own store
                    HF.gemm_split_up2(sw_up, skip.contiguous(), shift=shift, act=HF.ACT_RELU, out=signal[:, half:])
                    return signal
                if sw_up is not None:
                    y = HF.gemm_split(sw_up, skip.contiguous(), shift=shift, act=HF.ACT_RELU)
                else:
                    y = HF.affine_act_(torch.mm(wgt[:, :half], skip.view(half, sh * sw)).view(1, half, sh, sw), scale, shift, HF.ACT_RELU)
                pooled_const = False
            else:
                y = torch.mm(wgt, torch.cat((skip, t), dim=1).view(cin, sh * sw)).view(1, half, sh, sw)
                y = HF.affine_act_(y, scale, shift, HF.ACT_RELU)
            if level > 0:
                t = wm.upsample(y)
            else:
                # nearest 2x upsample straight into the right half of the signal
                signal[:, half:].view(half, sh, 2, sw, 2).copy_(y.view(half, sh, 1, sw, 1).expand(half, sh, 2, sw, 2))
        return signal


def _fuse_backbone(bb):
    """Swap every MBConv block, the head conv and the feature reducers of an EfficientNet for their fused forms, threading
    the deferred-BN-shift offsets (FusedMBConv docstring) from producer to consumers."""
    blocks = list(bb._blocks)
    offset, tap = None, 0
    bb._fused_fc = nn.ModuleDict()
    for idx, blk in enumerate(blocks):
        conv = blk._depthwise_conv
        ok = conv.kernel_size[0] in (3, 5) and conv.stride[0] in (1, 2) and isinstance(blk._bn1, nn.BatchNorm2d)
        tapped = bb._res_feat_mask[idx]
        fc = getattr(bb, f'_feat_fc_{tap}', None) if (tapped and bb.out_feat_scale is not None) else None
        if ok:
            nxt = blocks[idx + 1] if idx + 1 < len(blocks) else None
            nxt_ok = nxt is None or (nxt.expand != 1 and nxt._depthwise_conv.kernel_size[0] in (3, 5)
                                     and isinstance(nxt._bn1, nn.BatchNorm2d))
            # every consumer of the output must be a 1x1 conv we control: next block's expand (or the head), the reducer
            defer = nxt_ok and (not tapped or isinstance(fc, nn.Sequential))
            blk._fused_dw = FusedMBConv(blk, in_offset=offset, defer_shift=defer)
            offset = blk._fused_dw.out_offset
        else:
            assert offset is None
        if tapped:
            if isinstance(fc, nn.Sequential):
                fused = FusedPointwise(fc[0], fc[1], act=0)
                if offset is not None:
                    fused.absorb_input_offset(offset)
                bb._fused_fc[str(tap)] = fused
            else:
                assert offset is None
            tap += 1
    stem = bb._conv_stem
    if stem.kernel_size == (3, 3) and stem.stride == (2, 2) and stem.in_channels == 3 and isinstance(bb._bn0, nn.BatchNorm2d):
        bb._fused_stem = FusedStem(stem, bb._bn0)
        f0 = blocks[0]._fused_dw if blocks else None
        if f0 is not None and f0.expand is None and f0.k == 3 and f0.stride == 1 and not f0.skip:
            f0._stem = [bb._fused_stem]           # EfficientNet._extract_features_list then hands block 0 the image
    bb._fused_head = FusedPointwise(bb._conv_head, bb._bn1, act=3)
    if offset is not None:
        bb._fused_head.absorb_input_offset(offset)


def _install_fused(model):
    """(Re)build the fused routes of the encoder and the context head from the model's CURRENT parameters."""
    bb, wm = model.backbone, model.weight_mapper
    for blk in bb._blocks:
        blk._fused_dw = None
    bb._fused_stem = bb._fused_head = bb._fused_fc = None
    _fuse_backbone(bb)
    dev = next(model.parameters()).device
    bb.to(dev)                                      # new non-persistent buffers follow the model's device
    if type(wm).__name__ == 'WeightMapper' and hasattr(wm, 'in_conv') and hasattr(wm, 'up_blocks') and wm.levels >= 2:
        wm._fused = FusedContextHead(wm).to(dev)
        wm._fused.split_gemm = bool(getattr(model, '_hs_split_gemm', False))
    for m in bb.modules():                          # prepare_for_inference(split_gemm=...): our GEMM for the 1x1 convs
        if isinstance(m, FusedPointwise):
            m.split_gemm = bool(getattr(model, '_hs_split_gemm', False))


def set_ir_math(model, mode):
    """Arithmetic of the decoder's fused inverted-residual levels (include/hyperseg_hip.h, hs_ir_math): sets the ``ir_math``
    attribute every such module hands to its launches -- 'f32' (the module default: exact f32 matrix cores), 'split' or
    'auto' (f16 matrix cores on split operands, f32-class, where that form exists).  Returns the number of modules touched."""
    from .. import functional as HF
    if mode not in HF.IR_MATH:
        raise ValueError(f'ir math {mode!r}: expected one of {sorted(HF.IR_MATH)}')
    n = 0
    for m in model.modules():
        if type(m).__name__ == 'HyperPatchInvertedResidual':
            m.ir_math = mode
            n += 1
    return n


def prepare_for_inference(model, fold_bn=True, channels_last=False, fused_depthwise=False, split_gemm=False, ir_math='auto',
                          chain_k1=True):
    """In place; returns the number of BatchNorms folded by ``fold_bn``.  ``model``: a HyperGen in eval mode (module
    docstring for what each switch does).  The fused routes are installed first, so ``fold_bn`` only touches the
    Conv -> BatchNorm pairs that no fused route reads.  ``ir_math``: :func:`set_ir_math` for the decoder ('auto' = the
    f16-split inverted residual wherever it exists -- what serving and bench.py run; None leaves the modules' 'f32').
    ``chain_k1``: the decoder's three coarse k = 1 levels as ONE launch with in-launch neighbour hand-offs (hs_k1_chain_fwd;
    v1_0 decoders whose whole grid is resident at once -- the others keep one launch per level).  One frame in flight per model:
    the launch keeps its generation counter in a per-decoder workspace."""
    assert not model.training, 'call model.eval() first'
    folded = 0
    if ir_math is not None:
        set_ir_math(model, ir_math)
    if hasattr(model, 'decoder'):
        model.decoder.chain_k1 = bool(chain_k1)
    if fused_depthwise and any(getattr(b, '_fused_dw', None) is not None for b in model.backbone._blocks):
        raise RuntimeError('prepare_for_inference(fused_depthwise=True) was already applied to this model: the deferred '
                           'BatchNorm shifts would be absorbed twice')
    wm = model.weight_mapper
    if fused_depthwise:
        model._hs_split_gemm = bool(split_gemm)
        _install_fused(model)
        if not fold_bn and not getattr(model, '_hs_refresh_hook', None):
            # the folded BN affines and the deferred-shift chain are derived from the parameters at this moment: rebuild
            # them whenever a state dict is loaded afterwards (fold_bn rewrites parameters, so there the order is fixed)
            def _refresh(module, incompatible_keys):
                _install_fused(module)
            model._hs_refresh_hook = model.register_load_state_dict_post_hook(_refresh)
    if fold_bn:
        bb = model.backbone
        pairs = ([] if getattr(bb, '_fused_stem', None) is not None else [('_conv_stem', '_bn0')]) + \
                ([] if getattr(bb, '_fused_head', None) is not None else [('_conv_head', '_bn1')])
        folded += _fold_pairs(bb, pairs)
        for blk in bb._blocks:
            if blk._fused_dw is None:
                folded += _fold_pairs(blk, [('_expand_conv', '_bn0'), ('_project_conv', '_bn2'), ('_depthwise_conv', '_bn1')])
        fused_fc = getattr(bb, '_fused_fc', None)
        head_mods = [] if getattr(wm, '_fused', None) is not None else list(wm.named_modules())
        for name, m in list(bb.named_children()) + head_mods:
            if isinstance(m, nn.Sequential) and not (fused_fc is not None and name.startswith('_feat_fc_')):
                folded += _fold_sequential(m)
    if channels_last:
        model.backbone.to(memory_format=torch.channels_last)
        model.weight_mapper.to(memory_format=torch.channels_last)
    return folded


class GraphedModel(nn.Module):
    """Serving wrapper: ONE HIP-graph replay per forward instead of ~200 eager launches (HyperSeg-M: the eager launch
    path costs 2.7 ms of host time per frame, the replay 0.98 ms of GPU time).  The reference has no counterpart -- its
    FPS harness launches eagerly (hyperseg/test_fps.py:173-188); ``hyperseg_amd.fps --graph`` runs that harness' protocol
    through this wrapper.

    ``forward(x)``: ``x`` a single tensor, on the device or in (pinned) host memory -- it is copied into the graph's
    static input buffer on the current stream, so the host-to-device copy IS the staging copy.  One graph is captured per
    (shape, dtype) on first use (``warmup`` eager forwards on a side stream first: library handles, workspaces and the
    lazily built buffers of the fused routes must exist before capture).  The returned tensor is the graph's static output:
    valid until the next forward of the same shape (``clone_output=True`` hands out copies).  Anything the graph cannot
    serve takes the wrapped model's eager path: list inputs (pyramids), training mode, inputs or parameters that need a
    gradient, CPU models.  Parameters are read through their storage, so in-place updates are seen; ``load_state_dict``
    on the wrapped model (which rebuilds the fused routes' folded buffers) and ``reset()`` drop the captured graphs."""

    def __init__(self, model, masks=False, warmup=3, clone_output=False, max_graphs=8):
        super().__init__()
        self.model = model
        self.masks, self.warmup, self.clone_output, self.max_graphs = bool(masks), int(warmup), bool(clone_output), int(max_graphs)
        self._graphs = {}                      # (shape, dtype, device) -> (graph, static_in, static_out, chained-launch owners)
        self._replays = 0
        self._hook = model.register_load_state_dict_post_hook(lambda module, incompatible: self.reset())

    def reset(self):
        self._graphs.clear()

    def _eager(self, x):
        return self.model.segment(x) if self.masks else self.model(x)

    def _graphable(self, x):
        if not isinstance(x, torch.Tensor) or self.model.training or x.requires_grad:
            return False
        p = next(self.model.parameters(), None)
        if p is None or not p.is_cuda:
            return False
        return not (torch.is_grad_enabled() and any(q.requires_grad for q in self.model.parameters()))

    def _capture(self, key, x, device):
        if len(self._graphs) >= self.max_graphs:
            self._graphs.pop(next(iter(self._graphs)))           # oldest shape goes
        with torch.cuda.device(device), torch.no_grad():
            static_in = torch.empty(x.shape, dtype=x.dtype, device=device)
            static_in.copy_(x)
            main = torch.cuda.current_stream(device)
            side = torch.cuda.Stream(device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for _ in range(max(1, self.warmup)):
                    self._eager(static_in)
            main.wait_stream(side)
            torch.cuda.synchronize(device)
            graph = torch.cuda.CUDAGraph()
            from .. import functional as HF
            chained = HF.CHAIN_LAUNCHES_CAPTURED[0]
            with torch.cuda.graph(graph):
                static_out = self._eager(static_in)
            # a graph that contains the decoder's chained launch (hs_decoder_chain_fwd: its whole grid must be resident at once) is
            # replayed behind functional.ChainGate: never beside another chained launch on this device, whatever stream that one is on
            chains = [m._k1_chain for m in self.model.modules() if getattr(m, '_k1_chain', None) is not None] \
                if HF.CHAIN_LAUNCHES_CAPTURED[0] != chained else []
        self._graphs[key] = (graph, static_in, static_out, chains)
        return self._graphs[key]

    accepts_host_input = True                  # hyperseg_amd.fps.measure_fps hands the pinned host batch over as it is

    def forward(self, x):
        p = next(self.model.parameters(), None)
        if not self._graphable(x):
            if isinstance(x, torch.Tensor) and p is not None and x.device != p.device:
                x = x.to(p.device, non_blocking=True)
            return self._eager(x)
        device = p.device
        key = (tuple(x.shape), x.dtype, device)
        entry = self._graphs.get(key)
        if entry is None:
            entry = self._capture(key, x, device)
        graph, static_in, static_out, chains = entry
        with torch.cuda.device(device):        # the replay and the input copy go to the MODEL's device and its current stream
            if chains:
                from .. import functional as HF
                for ch in chains:              # pinned mirrors of the kernel's error word: a frame built on an abandoned wait raises
                    ch.check_errors()
                gate = HF.ChainGate.of(device)
                with gate.lock:
                    cur = gate.enter(device)
                    static_in.copy_(x, non_blocking=True)
                    graph.replay()
                    gate.leave(cur)
                self._replays += 1
                if self._replays % HF.K1Chain.POLL_EVERY == 0:
                    for ch in chains:
                        ch.request_error_copy(device)
            else:
                static_in.copy_(x, non_blocking=True)
                graph.replay()
            return static_out.clone() if self.clone_output else static_out
