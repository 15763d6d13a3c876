"""Training path of the dynamic patch-wise convolution: ``torch.autograd.Function`` wrappers around the HIP forward
and backward kernels (SURVEY.md section 8b "Autograd", Appendix E; BASELINE config 5).

The reference has no custom backward: autograd differentiates its ATen ops (F.pad, unfold, grouped conv2d, fold).
Here the convolution itself -- forward, per-patch weight gradient, input gradient -- runs in HIP kernels; the cheap glue
around it in TRAINING mode (stage-input concatenation, BatchNorm with batch statistics, activations, the grouped 1x1
``signal2weights`` convolution) stays stock PyTorch so that autograd composes the whole decoder.  Inference never comes
through this module: it uses the fused kernels (one launch per level).
"""

import os

import torch

from . import _hip
from . import functional as HF


def needs_grad(*tensors):
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


DTYPE_CODES = {torch.float32: 0, torch.bfloat16: 1}       # hs_dtype of include/hyperseg_hip.h


def _plain_conv(kind, dtype, a, b, ld, shape, meta, out):
    """One of the storage-typed training kernels (hs_patch_conv_plain_*): fp32, or bf16 storage + fp32 accumulation."""
    (fh, fw), c_out, k, pad, mode, groups = meta
    bsz, c_in, h, w = shape
    fn = getattr(_hip.lib, 'hs_patch_conv_plain_' + kind)
    code = DTYPE_CODES[dtype]
    if kind == 'bwd_w':
        st = fn(code, _hip.dev_ptr(a, 'x', dtype), _hip.dev_ptr(b, 'dy', dtype), bsz, c_in, h, w, fh, fw, c_out, k, pad,
                HF.PAD_MODES[mode], groups, out.data_ptr(), ld, _hip.stream_ptr())
    else:
        st = fn(code, _hip.dev_ptr(a, 'x' if kind == 'fwd' else 'dy', dtype), b.data_ptr(), ld, bsz, c_in, h, w, fh, fw,
                c_out, k, pad, HF.PAD_MODES[mode], groups, out.data_ptr(), _hip.stream_ptr())
    _hip.check(st, 'hs_patch_conv_plain_' + kind)
    return out


class _BankGradBuffer:
    """One gradient buffer for the three column ranges BankSlices hands out (round 5): the layers that consume the ranges write their
    weight gradients straight into views of ONE (patches, ld) tensor, and BankSlices.backward returns it as it is -- where each layer
    allocated its own (patches, range) tensor, the backward of every inverted-residual level was a concatenation launch over the whole
    bank (two CatArrayBatchedCopy per config-5 step).  Created per BankSlices.forward, filled per backward pass, dropped when returned.

    Contract: ONE consumer per range and backward pass.  A range handed to two consumers (not something this package's modules do) would
    have the second weight-gradient kernel overwrite the first's result inside the shared buffer before autograd adds the two -- so a
    SECOND request for a range in the same pass returns None and the caller allocates privately; ``owns`` then fails for the
    shared views' sum and BankSlices.backward takes the concatenating route."""

    def __init__(self, shape, ranges):
        self.shape, self.ranges, self.buf = tuple(shape), tuple(ranges), None
        self.taken = set()

    def view(self, index, device):
        if index in self.taken:
            return None                                          # a second consumer of the same range: private allocation, autograd adds
        self.taken.add(index)
        if self.buf is None:
            self.buf = torch.empty(self.shape, device=device, dtype=torch.float32)
            if self.shape[1] > self.ranges[-1][1]:
                self.buf[:, self.ranges[-1][1]:].zero_()        # the row's pad columns (read by nobody; kept finite)
        c0, c1 = self.ranges[index]
        return self.buf[:, c0:c1]

    def owns(self, grads):
        if self.buf is None or any(g is None for g in grads):
            return False
        return all(g.dtype == torch.float32 and g.stride() == self.buf.stride() and g.shape == (self.shape[0], c1 - c0)
                   and g.data_ptr() == self.buf.data_ptr() + 4 * c0 for g, (c0, c1) in zip(grads, self.ranges))


def _grad_slot(bank):
    """(buffer, index) a BankSlices range carries, or None (read in forward, from the tensor object the caller passed)."""
    return getattr(bank, '_hs_grad_slot', None)


def _dbank_for(slot, bank, written_cols, device):
    """The weight-gradient tensor of a layer whose kernels write columns [0, written_cols) of every row: a view of the shared buffer when
    the layer's bank is a BankSlices range of exactly that width, else its own allocation (zero-filled where the kernels do not write)."""
    if slot is not None and written_cols == bank.shape[1]:
        shared = slot[0].view(slot[1], device)
        if shared is not None:
            return shared
    alloc = torch.empty if written_cols == bank.shape[1] else torch.zeros
    return alloc(bank.shape[0], bank.shape[1], device=device, dtype=torch.float32)


class PatchConv(torch.autograd.Function):
    """y = patch_conv(x, bank): Op A / Op B with plain tensors.  Saves x and the bank; backward launches the input- and
    the per-patch weight-gradient kernels.  fp32 tensors take the fp32 kernels; under ``torch.autocast('cuda',
    dtype=torch.bfloat16)`` (or with a bf16 ``x``) the ACTIVATIONS and their gradients are stored as bf16 and every sum is
    accumulated in fp32 (hs_patch_conv_plain_*: BASELINE config 5).  The bank and its gradient stay fp32 in both cases: they
    sit between this layer and signal2weights, which is fp32, so a bf16 bank meant one cast launch per layer and direction
    (30 of the 147 launches of the config-5 step under autocast, visit r4m) for a few per cent of a launch's bytes."""

    @staticmethod
    def forward(ctx, x, bank, grid, c_out, k, padding, padding_mode, groups):
        # (no custom_fwd(cast_inputs=...): it would narrow the bank too, and widen its gradient again in backward)
        if torch.is_autocast_enabled('cuda') and x.is_cuda and x.is_floating_point():
            x = x.to(torch.bfloat16)
        if x.dtype not in DTYPE_CODES:
            raise NotImplementedError(f'PatchConv: dtype {x.dtype} is not supported (supported: '
                                      f'{", ".join(str(d) for d in DTYPE_CODES)})')
        x = x.contiguous()
        ctx.grad_slot = _grad_slot(bank)
        if bank.dtype != torch.float32:
            bank = bank.float()
        if bank.stride(1) != 1:
            bank = bank.contiguous()
        meta = (tuple(grid), c_out, k, padding, padding_mode, groups)
        if x.dtype == torch.float32:
            y = HF.patch_conv(x, grid, bank, c_out, k, padding, padding_mode, groups)
        else:
            with _hip.device_scope(x.device):
                y = _plain_conv('fwd', x.dtype, x, bank, bank.stride(0), x.shape, meta,
                                torch.empty(x.shape[0], c_out, x.shape[2], x.shape[3], device=x.device, dtype=x.dtype))
        ctx.save_for_backward(x, bank)
        ctx.meta = meta
        return y

    @staticmethod
    def backward(ctx, dy):
        x, bank = ctx.saved_tensors
        (fh, fw), c_out, k, pad, mode, groups = ctx.meta
        dy = dy.contiguous().to(x.dtype)
        b, c_in, h, w = x.shape
        dx = dbank = None
        with _hip.device_scope(x.device):
            stream = _hip.stream_ptr()
            bank_ptr, ld = bank.data_ptr(), bank.stride(0)
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                if x.dtype == torch.float32:
                    st = _hip.lib.hs_patch_conv_bwd_input(_hip.dev_ptr(dy, 'dy'), bank_ptr, ld, b, c_in, h, w, fh, fw, c_out,
                                                          k, pad, HF.PAD_MODES[mode], groups, dx.data_ptr(), stream)
                    _hip.check(st, 'hs_patch_conv_bwd_input')
                else:
                    _plain_conv('bwd_in', x.dtype, dy, bank, ld, x.shape, ctx.meta, dx)
            if ctx.needs_input_grad[1]:
                rows = c_out * (c_in // groups) * k * k
                # the kernels write columns [0, rows) of every patch row; only trailing pad columns need the zero fill
                full = _dbank_for(ctx.grad_slot, bank, rows, x.device)
                assert rows <= full.shape[1]
                if x.dtype == torch.float32:
                    st = _hip.lib.hs_patch_conv_bwd_weight(_hip.dev_ptr(x, 'x'), _hip.dev_ptr(dy, 'dy'), b, c_in, h, w, fh, fw,
                                                           c_out, k, pad, HF.PAD_MODES[mode], groups, full.data_ptr(),
                                                           full.stride(0), stream)
                    _hip.check(st, 'hs_patch_conv_bwd_weight')
                else:
                    _plain_conv('bwd_w', x.dtype, x, dy, full.stride(0), x.shape, ctx.meta, full)
                dbank = full
        return dx, dbank, None, None, None, None, None, None


def _tile_call(name, dtype, src, b, c, h, w, grid, dst, layout=None):
    args = (DTYPE_CODES[dtype], src.data_ptr(), b, c, h, w, grid[0], grid[1], dst.data_ptr())
    st = getattr(_hip.lib, name)(*args, _hip.stream_ptr()) if layout is None else getattr(_hip.lib, name)(*args, int(layout), _hip.stream_ptr())
    _hip.check(st, name)
    return dst


class HaloTiles(torch.autograd.Function):
    """x (B, C, H, W) -> its reflect-padded halo tiles: F.pad(reflect) -> unfold -> unfold -> permute -> reshape of models/hyperseg_v1_0.py
    _run_train as ONE gather (hs_halo_tiles_fwd), the adjoint as one gather too.  ``patch_major`` False: the tiles side by side as one
    image (B, C, fh (ph+2), fw (pw+2)); True: one tile after the other, (B fh fw, C, ph+2, pw+2) -- every operand of a patch contiguous,
    the block's 1x1 layers then run as patch convolutions with a (1, 1) grid over B fh fw frames."""

    @staticmethod
    def forward(ctx, x, grid, patch_major=False):
        x = x.contiguous()
        b, c, h, w = x.shape
        fh, fw = grid
        ctx.meta = (b, c, h, w, (fh, fw), x.dtype, bool(patch_major))
        with _hip.device_scope(x.device):
            shape = (b * fh * fw, c, h // fh + 2, w // fw + 2) if patch_major else (b, c, fh * (h // fh + 2), fw * (w // fw + 2))
            out = torch.empty(shape, device=x.device, dtype=x.dtype)
            return _tile_call('hs_halo_tiles_fwd', x.dtype, x, b, c, h, w, (fh, fw), out, layout=patch_major)

    @staticmethod
    def backward(ctx, dt):
        b, c, h, w, grid, dtype, pm = ctx.meta
        dt = dt.contiguous().to(dtype)
        with _hip.device_scope(dt.device):
            return _tile_call('hs_halo_tiles_bwd', dtype, dt, b, c, h, w, grid, torch.empty(b, c, h, w, device=dt.device, dtype=dtype),
                              layout=pm), None, None


class TileInterior(torch.autograd.Function):
    """The image of halo tiles -> (B, C, H, W) without the halos (hs_tile_interior_fwd); adjoint: zeros on the halos."""

    @staticmethod
    def forward(ctx, t, size, grid):
        t = t.contiguous()
        b, c = t.shape[:2]
        h, w = size
        ctx.meta = (b, c, h, w, tuple(grid), t.dtype, tuple(t.shape))
        with _hip.device_scope(t.device):
            return _tile_call('hs_tile_interior_fwd', t.dtype, t, b, c, h, w, grid, torch.empty(b, c, h, w, device=t.device, dtype=t.dtype))

    @staticmethod
    def backward(ctx, dy):
        b, c, h, w, grid, dtype, shape = ctx.meta
        dy = dy.contiguous().to(dtype)
        with _hip.device_scope(dy.device):
            return _tile_call('hs_tile_interior_bwd', dtype, dy, b, c, h, w, grid, torch.empty(shape, device=dy.device, dtype=dtype)), None, None


def _dw_tiles_input_gradient(t, dy, bank, shape, grid, patch_major):
    """hs_dw_tiles_bwd_in: the valid depthwise adjoint, a halo-tile image like ``t`` (same layout and storage type)."""
    b, c, h, w = shape
    dt = torch.empty_like(t)
    st = _hip.lib.hs_dw_tiles_bwd_in(DTYPE_CODES[t.dtype], dy.data_ptr(), bank.data_ptr(), bank.stride(0), b, c, h, w, grid[0], grid[1],
                                     dt.data_ptr(), int(patch_major), _hip.stream_ptr())
    _hip.check(st, 'hs_dw_tiles_bwd_in')
    return dt


class DwTilesValid(torch.autograd.Function):
    """The middle layer of a train-mode v1_0 inverted residual: a VALID depthwise 3x3 of every halo tile with the patch's own taps
    (hyperseg_v1_0.py:352-360), tiles -> (B, C, H, W), one launch per direction and operand (hs_dw_tiles_fwd / _bwd_in / _bwd_w).
    ``t``: the tiles as HaloTiles lays them out (``patch_major`` alike); ``bank``: the depthwise column range (P, 9 C) of the block's
    fp32 bank (a view)."""

    @staticmethod
    def forward(ctx, t, bank, size, grid, patch_major=False):
        ctx.grad_slot = _grad_slot(bank)
        t = t.contiguous()
        h, w = size
        c = t.shape[1]
        b = t.shape[0] // (grid[0] * grid[1]) if patch_major else t.shape[0]
        if bank.dtype != torch.float32 or bank.stride(1) != 1:
            bank = bank.float().contiguous()
        ctx.meta = (b, c, h, w, tuple(grid), t.dtype, bool(patch_major))
        with _hip.device_scope(t.device):
            y = torch.empty(b, c, h, w, device=t.device, dtype=t.dtype)
            st = _hip.lib.hs_dw_tiles_fwd(DTYPE_CODES[t.dtype], t.data_ptr(), bank.data_ptr(), bank.stride(0), b, c, h, w, grid[0], grid[1],
                                          y.data_ptr(), int(patch_major), _hip.stream_ptr())
            _hip.check(st, 'hs_dw_tiles_fwd')
        ctx.save_for_backward(t, bank)
        return y

    @staticmethod
    def backward(ctx, dy):
        t, bank = ctx.saved_tensors
        b, c, h, w, grid, dtype, pm = ctx.meta
        dy = dy.contiguous().to(dtype)
        dt = dbank = None
        with _hip.device_scope(dy.device):
            if ctx.needs_input_grad[0]:
                dt = _dw_tiles_input_gradient(t, dy, bank, (b, c, h, w), grid, pm)
            if ctx.needs_input_grad[1]:
                dbank = _dbank_for(ctx.grad_slot, bank, 9 * c, dy.device)
                st = _hip.lib.hs_dw_tiles_bwd_w(DTYPE_CODES[dtype], t.data_ptr(), dy.data_ptr(), b, c, h, w, grid[0], grid[1], dbank.data_ptr(),
                                                dbank.stride(0), int(pm), _hip.stream_ptr())
                _hip.check(st, 'hs_dw_tiles_bwd_w')
        return dt, dbank, None, None, None


def dw_tiles_supported(t, size, grid):
    """hs_dw_tiles_*: CUDA fp32 / bf16, even patch width (two adjacent elements per thread)."""
    return tiles_supported(t) and USE_HIP_DW_TILES and size[0] % grid[0] == 0 and size[1] % grid[1] == 0 and (size[1] // grid[1]) % 2 == 0


USE_HIP_DW_TILES = True     # tests switch it off to compare with the two-launch route (zero-padded depthwise on the tile image + TileInterior)
USE_PATCH_MAJOR_TILES = True   # the block's tiles one after the other instead of side by side (contiguous per-patch operands); off: the image of tiles


def tiles_supported(x):
    return x.is_cuda and x.dtype in DTYPE_CODES and x.shape[0] * x.shape[1] <= 65535 and x.shape[2] >= 2 and x.shape[3] >= 2


class BNActTrain(torch.autograd.Function):
    """BatchNorm2d (training mode, batch statistics, running estimates updated in place) + none / ReLU / ReLU6 in two launches per
    direction (hs_bn_act_train_fwd / _bwd).  fp32 parameters; x in fp32 or bf16 storage."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, act, counter=None):
        x = x.contiguous()
        b, c = x.shape[:2]
        px = x.numel() // (b * c)
        dev = x.device
        with _hip.device_scope(dev):
            y = torch.empty_like(x)
            mean, invstd = torch.empty(c, device=dev, dtype=torch.float32), torch.empty(c, device=dev, dtype=torch.float32)
            ws = torch.empty(int(_hip.lib.hs_bn_train_workspace(c)), device=dev, dtype=torch.uint8)
            st = _hip.lib.hs_bn_act_train_fwd(DTYPE_CODES[x.dtype], x.data_ptr(), b, c, px,
                                              weight.data_ptr() if weight is not None else None,
                                              bias.data_ptr() if bias is not None else None,
                                              running_mean.data_ptr() if running_mean is not None else None,
                                              running_var.data_ptr() if running_var is not None else None,
                                              float(momentum), float(eps), int(act), mean.data_ptr(), invstd.data_ptr(), ws.data_ptr(),
                                              y.data_ptr(), counter.data_ptr() if counter is not None else None, _hip.stream_ptr())
            _hip.check(st, 'hs_bn_act_train_fwd')
        ctx.save_for_backward(x, weight, bias, mean, invstd)
        ctx.meta = (b, c, px, float(eps), int(act))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias, mean, invstd = ctx.saved_tensors
        b, c, px, eps, act = ctx.meta
        dy = dy.contiguous().to(x.dtype)
        dev = x.device
        with _hip.device_scope(dev):
            dx = torch.empty_like(x)
            dg = torch.empty(c, device=dev, dtype=torch.float32) if weight is not None else None
            db = torch.empty(c, device=dev, dtype=torch.float32) if bias is not None else None
            ws = torch.empty(int(_hip.lib.hs_bn_train_workspace(c)), device=dev, dtype=torch.uint8)
            st = _hip.lib.hs_bn_act_train_bwd(DTYPE_CODES[x.dtype], x.data_ptr(), dy.data_ptr(), b, c, px,
                                              weight.data_ptr() if weight is not None else None,
                                              bias.data_ptr() if bias is not None else None, mean.data_ptr(), invstd.data_ptr(), eps, act,
                                              ws.data_ptr(), dx.data_ptr(), dg.data_ptr() if dg is not None else None,
                                              db.data_ptr() if db is not None else None, _hip.stream_ptr())
            _hip.check(st, 'hs_bn_act_train_bwd')
        return dx, dg, db, None, None, None, None, None, None


def _bn_hip_act(bn, act_layer, x):
    """Activation code (0 none, 1 ReLU, 2 ReLU6) when ``act_layer(bn(x))`` can go through the fused training kernels -- a plain
    BatchNorm2d in training mode with fp32 parameters on x's device, batch statistics over more than one element -- else -1."""
    import torch.nn as nn
    act = 0 if act_layer is None else 1 if type(act_layer) is nn.ReLU else 2 if type(act_layer) is nn.ReLU6 else -1
    ok = (USE_HIP_BN and type(bn) is nn.BatchNorm2d and bn.training and bn.track_running_stats and bn.momentum is not None and act >= 0
          and x.is_cuda and x.dtype in DTYPE_CODES and x.dim() == 4 and bn.affine and bn.weight.dtype == torch.float32
          and x.shape[0] * x.shape[2] * x.shape[3] > 1 and x.shape[0] * x.shape[2] * x.shape[3] < 2 ** 31
          # raw data_ptrs of these go to the kernel: they must live on x's device (stock BN raises for a mismatch, and so does
          # the stock route)
          and bn.weight.device == x.device and bn.bias.device == x.device
          and bn.running_mean.device == x.device and bn.running_var.device == x.device)
    return act if ok else -1


def bn_act(bn, act_layer, x):
    """``act_layer(bn(x))`` -- through the fused training kernels when ``bn`` is a plain BatchNorm2d in training mode with fp32
    parameters on the GPU and the activation is None / ReLU / ReLU6; the stock modules otherwise (eval mode, other layers, CPU)."""
    act = _bn_hip_act(bn, act_layer, x)
    if act >= 0:
        nbt = bn.num_batches_tracked
        in_kernel = nbt is not None and nbt.device == x.device and nbt.dtype == torch.int64
        # the step counter is incremented by the kernel itself (one launch less per BatchNorm and step)
        y = BNActTrain.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, act, nbt if in_kernel else None)
        if nbt is not None and not in_kernel:
            nbt.add_(1)
        HF.bump_weights_epoch()               # running statistics changed through a raw pointer: no tensor version moved
        return y
    if bn.training:
        HF.bump_weights_epoch()               # stock BN's running-statistics update does not bump their versions either
    y = bn(x)
    return y if act_layer is None else act_layer(y)


class DwTilesBN(torch.autograd.Function):
    """BatchNorm (training mode) + activation + the valid depthwise 3 x 3 of every halo tile WITHOUT the normalised copy of the tiles
    (round 5; hyperseg_v1_0.py:346-360): forward = hs_bn_train_stats_fwd on the raw tiles, then hs_dw_tiles_bn_fwd (normalise on load,
    statistics finalised and the running estimates updated by the same launch) -- two launches where BNActTrain + DwTilesValid took
    three, and the largest tensor of a config-5 step (the normalised tile image of level 4: 37 MB written and read back) never exists.
    Backward: the tap gradient from the raw tiles + saved statistics (hs_dw_tiles_bn_bwd_w), the tiles' gradient through the valid
    depthwise adjoint and then BatchNorm's own adjoint (hs_dw_tiles_bwd_in, hs_bn_act_train_bwd): the same four launches as before.
    Same arithmetic per value as the two Functions it replaces; bf16 storage skips the intermediate's rounding."""

    @staticmethod
    def forward(ctx, t, weight, bias, running_mean, running_var, momentum, eps, act, counter, bank, size, grid, patch_major):
        ctx.grad_slot = _grad_slot(bank)
        t = t.contiguous()
        h, w = size
        c = t.shape[1]
        fh, fw = grid
        b = t.shape[0] // (fh * fw) if patch_major else t.shape[0]
        px = t.numel() // (t.shape[0] * c)
        if bank.dtype != torch.float32 or bank.stride(1) != 1:
            bank = bank.float().contiguous()
        dev = t.device
        with _hip.device_scope(dev):
            y = torch.empty(b, c, h, w, device=dev, dtype=t.dtype)
            mean, invstd = torch.empty(c, device=dev, dtype=torch.float32), torch.empty(c, device=dev, dtype=torch.float32)
            ws = torch.empty(int(_hip.lib.hs_bn_train_workspace(c)), device=dev, dtype=torch.uint8)
            code = DTYPE_CODES[t.dtype]
            _hip.check(_hip.lib.hs_bn_train_stats_fwd(code, t.data_ptr(), t.shape[0], c, px, ws.data_ptr(), _hip.stream_ptr()), 'hs_bn_train_stats_fwd')
            st = _hip.lib.hs_dw_tiles_bn_fwd(code, t.data_ptr(), ws.data_ptr(), weight.data_ptr(), bias.data_ptr(),
                                             running_mean.data_ptr() if running_mean is not None else None,
                                             running_var.data_ptr() if running_var is not None else None, float(momentum), float(eps), int(act),
                                             mean.data_ptr(), invstd.data_ptr(), counter.data_ptr() if counter is not None else None,
                                             bank.data_ptr(), bank.stride(0), b, c, h, w, fh, fw, y.data_ptr(), int(patch_major), _hip.stream_ptr())
            _hip.check(st, 'hs_dw_tiles_bn_fwd')
        ctx.save_for_backward(t, weight, bias, mean, invstd, bank)
        ctx.meta = (b, c, h, w, (fh, fw), float(eps), int(act), bool(patch_major), px)
        return y

    @staticmethod
    def backward(ctx, dy):
        t, weight, bias, mean, invstd, bank = ctx.saved_tensors
        b, c, h, w, (fh, fw), eps, act, pm, px = ctx.meta
        dy = dy.contiguous().to(t.dtype)
        dev = t.device
        code = DTYPE_CODES[t.dtype]
        dt = dbank = dg = db = None
        with _hip.device_scope(dev):
            stream = _hip.stream_ptr()
            if ctx.needs_input_grad[9]:
                dbank = _dbank_for(ctx.grad_slot, bank, 9 * c, dev)
                st = _hip.lib.hs_dw_tiles_bn_bwd_w(code, t.data_ptr(), dy.data_ptr(), weight.data_ptr(), bias.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                                   act, b, c, h, w, fh, fw, dbank.data_ptr(), dbank.stride(0), int(pm), stream)
                _hip.check(st, 'hs_dw_tiles_bn_bwd_w')
            if ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
                dt = torch.empty_like(t)
                dg = torch.empty(c, device=dev, dtype=torch.float32)
                db = torch.empty(c, device=dev, dtype=torch.float32)
                if USE_DW_BN_BWD_FUSED:
                    # round 6: the depthwise adjoint leaves BatchNorm1's two sums as one pair per workgroup; no statistics launch
                    npart = int(_hip.lib.hs_dw_tiles_bn_bwd_in_partials(b, h, w, fh, fw))
                    da = torch.empty_like(t)
                    part = torch.empty(c * npart * 2, device=dev, dtype=torch.float32)
                    st = _hip.lib.hs_dw_tiles_bn_bwd_in(code, dy.data_ptr(), bank.data_ptr(), bank.stride(0), t.data_ptr(), weight.data_ptr(), bias.data_ptr(),
                                                        mean.data_ptr(), invstd.data_ptr(), act, b, c, h, w, fh, fw, da.data_ptr(), part.data_ptr(), int(pm), stream)
                    _hip.check(st, 'hs_dw_tiles_bn_bwd_in')
                    st = _hip.lib.hs_bn_act_train_bwd_apply(code, t.data_ptr(), da.data_ptr(), t.shape[0], c, px, weight.data_ptr(), bias.data_ptr(),
                                                            mean.data_ptr(), invstd.data_ptr(), act, part.data_ptr(), npart, dt.data_ptr(), dg.data_ptr(),
                                                            db.data_ptr(), stream)
                    _hip.check(st, 'hs_bn_act_train_bwd_apply')
                else:
                    da = _dw_tiles_input_gradient(t, dy, bank, (b, c, h, w), (fh, fw), pm)   # gradient of the (never materialised) normalised tiles, stored in t's type
                    ws = torch.empty(int(_hip.lib.hs_bn_train_workspace(c)), device=dev, dtype=torch.uint8)
                    st = _hip.lib.hs_bn_act_train_bwd(code, t.data_ptr(), da.data_ptr(), t.shape[0], c, px, weight.data_ptr(), bias.data_ptr(),
                                                      mean.data_ptr(), invstd.data_ptr(), eps, act, ws.data_ptr(), dt.data_ptr(), dg.data_ptr(), db.data_ptr(), stream)
                    _hip.check(st, 'hs_bn_act_train_bwd')
        return dt, dg, db, None, None, None, None, None, None, dbank, None, None, None


def dw_tiles_bn(bn, act_layer, t, bank, size, grid, patch_major):
    """``DwTilesValid(bn_act(bn, act_layer, t), bank)`` as DwTilesBN where the fused training BatchNorm applies (and USE_DW_BN_FUSED),
    the two Functions otherwise."""
    act = _bn_hip_act(bn, act_layer, t) if USE_DW_BN_FUSED and getattr(bn, 'weight', None) is not None else -1      # nn.Identity / 'Unit' norms: no weight -> the two Functions
    if act < 0:
        return DwTilesValid.apply(bn_act(bn, act_layer, t), bank, size, grid, patch_major)
    nbt = bn.num_batches_tracked
    in_kernel = nbt is not None and nbt.device == t.device and nbt.dtype == torch.int64
    y = DwTilesBN.apply(t, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, act, nbt if in_kernel else None,
                        bank, tuple(size), tuple(grid), bool(patch_major))
    if nbt is not None and not in_kernel:
        nbt.add_(1)
    HF.bump_weights_epoch()
    return y


USE_DW_BN_FUSED = True  # tests switch it off to compare with BNActTrain + DwTilesValid
USE_DW_BN_BWD_FUSED = True        # False: BatchNorm1's adjoint with its own statistics launch (hs_bn_act_train_bwd; bit-equal to the two Functions)
USE_SHARED_BANK_GRAD = True  # BankSlices: one gradient buffer for the three ranges (tests switch it off to compare with the concatenation)


class PatchConvBN(torch.autograd.Function):
    """BatchNorm (training mode) + activation + the k = 1 per-patch convolution that follows, WITHOUT the normalised copy of the input
    (round 5; BatchNorm2 + ReLU6 + the last 1 x 1 layer of a train-mode inverted residual, hyperseg_v1_0.py:361-370): forward =
    hs_bn_train_stats_fwd on the raw hidden map, then hs_patch_conv_bn_fwd (statistics finalised, running estimates updated and the
    input normalised as it is read) -- two launches where BNActTrain + PatchConv took three.  Backward: the weight gradient from the raw
    input + saved statistics (hs_patch_conv_bn_bwd_w), the input gradient through the convolution's adjoint and then BatchNorm's own
    (hs_patch_conv_plain_bwd_in, hs_bn_act_train_bwd).  Same arithmetic per value as the two Functions it replaces."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, act, counter, bank, grid, c_out):
        ctx.grad_slot = _grad_slot(bank)
        if torch.is_autocast_enabled('cuda') and x.is_floating_point():
            x = x.to(torch.bfloat16)
        x = x.contiguous()
        b, c, h, w = x.shape
        fh, fw = grid
        if bank.dtype != torch.float32 or bank.stride(1) != 1:
            bank = bank.float().contiguous()
        dev = x.device
        with _hip.device_scope(dev):
            y = torch.empty(b, c_out, h, w, device=dev, dtype=x.dtype)
            mean, invstd = torch.empty(c, device=dev, dtype=torch.float32), torch.empty(c, device=dev, dtype=torch.float32)
            ws = torch.empty(int(_hip.lib.hs_bn_train_workspace(c)), device=dev, dtype=torch.uint8)
            code = DTYPE_CODES[x.dtype]
            _hip.check(_hip.lib.hs_bn_train_stats_fwd(code, x.data_ptr(), b, c, h * w, ws.data_ptr(), _hip.stream_ptr()), 'hs_bn_train_stats_fwd')
            st = _hip.lib.hs_patch_conv_bn_fwd(code, x.data_ptr(), ws.data_ptr(), weight.data_ptr(), bias.data_ptr(),
                                               running_mean.data_ptr() if running_mean is not None else None,
                                               running_var.data_ptr() if running_var is not None else None, float(momentum), float(eps), int(act),
                                               mean.data_ptr(), invstd.data_ptr(), counter.data_ptr() if counter is not None else None,
                                               bank.data_ptr(), bank.stride(0), b, c, h, w, fh, fw, int(c_out), y.data_ptr(), _hip.stream_ptr())
            _hip.check(st, 'hs_patch_conv_bn_fwd')
        ctx.save_for_backward(x, weight, bias, mean, invstd, bank)
        ctx.meta = (b, c, h, w, (fh, fw), float(eps), int(act), int(c_out))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias, mean, invstd, bank = ctx.saved_tensors
        b, c, h, w, (fh, fw), eps, act, c_out = ctx.meta
        dy = dy.contiguous().to(x.dtype)
        dev = x.device
        code = DTYPE_CODES[x.dtype]
        dx = dbank = dg = db = None
        with _hip.device_scope(dev):
            stream = _hip.stream_ptr()
            if ctx.needs_input_grad[9]:
                dbank = _dbank_for(ctx.grad_slot, bank, c_out * c, dev)
                st = _hip.lib.hs_patch_conv_bn_bwd_w(code, x.data_ptr(), dy.data_ptr(), weight.data_ptr(), bias.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                                     act, b, c, h, w, fh, fw, c_out, dbank.data_ptr(), dbank.stride(0), stream)
                _hip.check(st, 'hs_patch_conv_bn_bwd_w')
            if ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
                da = _conv_input_gradient(x, dy, bank, (fh, fw), c_out)              # gradient of the (never materialised) normalised map, in x's type
                dx = torch.empty_like(x)
                dg = torch.empty(c, device=dev, dtype=torch.float32)
                db = torch.empty(c, device=dev, dtype=torch.float32)
                ws = torch.empty(int(_hip.lib.hs_bn_train_workspace(c)), device=dev, dtype=torch.uint8)
                st = _hip.lib.hs_bn_act_train_bwd(code, x.data_ptr(), da.data_ptr(), b, c, h * w, weight.data_ptr(), bias.data_ptr(),
                                                  mean.data_ptr(), invstd.data_ptr(), eps, act, ws.data_ptr(), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), stream)
                _hip.check(st, 'hs_bn_act_train_bwd')
        return dx, dg, db, None, None, None, None, None, None, dbank, None, None


def _conv_input_gradient(x, dy, bank, grid, c_out):
    """hs_patch_conv_plain_bwd_in of a k = 1 layer: d input, a map like ``x`` (same storage type)."""
    return _plain_conv('bwd_in', x.dtype, dy, bank, bank.stride(0), x.shape, (tuple(grid), c_out, 1, 0, 'zeros', 1), torch.empty_like(x))


def patch_conv_bn(bn, act_layer, x, bank, grid, c_out):
    """``patch_conv_apply(bn_act(bn, act_layer, x), bank, grid, c_out, 1, 0, 'zeros', 1)`` as PatchConvBN where the fused training
    BatchNorm applies, the layer is inside hs_patch_conv_bn_fwd's range (c_out <= 32, <= 64 input channels, patches of >= 64 pixels)
    and USE_CONV_BN_FUSED; the two Functions otherwise."""
    act = _bn_hip_act(bn, act_layer, x) if USE_CONV_BN_FUSED and getattr(bn, 'weight', None) is not None else -1
    b, c, h, w = x.shape
    fh, fw = grid
    covered = act >= 0 and c_out <= 32 and c <= 64 and h % fh == 0 and w % fw == 0 and (h // fh) * (w // fw) >= 64 \
        and (not torch.is_autocast_enabled('cuda') or torch.get_autocast_dtype('cuda') == torch.bfloat16)
    if not covered:
        return patch_conv_apply(bn_act(bn, act_layer, x), bank, grid, c_out, 1, 0, 'zeros', 1)
    nbt = bn.num_batches_tracked
    in_kernel = nbt is not None and nbt.device == x.device and nbt.dtype == torch.int64
    y = PatchConvBN.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, act, nbt if in_kernel else None,
                          bank, tuple(grid), int(c_out))
    if nbt is not None and not in_kernel:
        nbt.add_(1)
    HF.bump_weights_epoch()
    return y


USE_CONV_BN_FUSED = True  # tests switch it off to compare with BNActTrain + PatchConv


USE_HIP_BN = True       # tests switch it off to compare with the stock modules



class S2WBanksTrain(torch.autograd.Function):
    """Every level's filter bank of a decoder from the signal, differentiably, one launch forward and three backward
    (hs_s2w_train_fwd / _bwd): ``apply(meta, signal, *weights)`` -> one patch-major bank (B fh fw, ld) per layer.  ``meta``: one dict
    per layer with signal_index, signal_channels, groups, rows; ``weights``: the layers' Conv2d weights (wc, K, 1, 1) as they are.
    Replaces, per level and step: the grouped 1x1 conv as a strided-batched GEMM (+ its two adjoint GEMMs), the re-layout into
    per-patch weights and its adjoint (BankPack), and autograd's zero fills / accumulations of the d signal slices."""

    @staticmethod
    def _table(meta, signal, weights, banks=None, dbanks=None, dws=None, dss=None):
        arr = (_hip.S2wTrainLayerC * len(meta))()
        for i, (m, w) in enumerate(zip(meta, weights)):
            a = arr[i]
            a.signal_index, a.signal_channels, a.groups = m['signal_index'], m['signal_channels'], m['groups']
            a.w, a.wc, a.rows = w.data_ptr(), w.shape[0], m['rows']
            a.ld = m['ld']
            a.bank = banks[i].data_ptr() if banks is not None else None
            a.dbank = dbanks[i].data_ptr() if dbanks is not None and dbanks[i] is not None else None
            a.dw = dws[i].data_ptr() if dws is not None and dws[i] is not None else None
            a.ds = dss[i].data_ptr() if dss is not None and dss[i] is not None else None
        return arr

    @staticmethod
    def forward(ctx, meta, signal, *weights):
        signal = signal.contiguous().float()
        weights = [w.contiguous().float() for w in weights]
        b, c, fh, fw = signal.shape
        p = b * fh * fw
        meta = [dict(m, ld=HF._round_up(m['rows'], 4)) for m in meta]
        with _hip.device_scope(signal.device):
            buf = torch.empty(p * sum(m['ld'] for m in meta), device=signal.device, dtype=torch.float32)
            banks, off = [], 0
            for m in meta:
                banks.append(buf[off:off + p * m['ld']].view(p, m['ld']))
                off += p * m['ld']
            arr = S2WBanksTrain._table(meta, signal, weights, banks=banks)
            st = _hip.lib.hs_s2w_train_fwd(signal.data_ptr(), b, c, fh, fw, arr, len(meta), _hip.stream_ptr())
            _hip.check(st, 'hs_s2w_train_fwd')
        ctx.save_for_backward(signal, *weights)
        ctx.meta = meta
        return tuple(banks)

    @staticmethod
    def backward(ctx, *dbanks):
        signal, *weights = ctx.saved_tensors
        meta = ctx.meta
        b, c, fh, fw = signal.shape
        p = b * fh * fw
        need_s = ctx.needs_input_grad[1]
        need_w = [ctx.needs_input_grad[2 + i] for i in range(len(meta))]
        dbs = []
        for m, g in zip(meta, dbanks):
            if g is not None and not (g.dtype == torch.float32 and g.stride(1) == 1 and g.stride(0) == m['ld'] and g.shape == (p, m['ld'])):
                g = g.float().contiguous()
            dbs.append(g)
        with _hip.device_scope(signal.device):
            dws = [torch.empty_like(w) if nw else None for w, nw in zip(weights, need_w)]
            dss = [torch.empty(b, m['signal_channels'], fh, fw, device=signal.device, dtype=torch.float32) if need_s else None for m in meta]
            dsig = torch.empty_like(signal) if need_s else None
            arr = S2WBanksTrain._table(meta, signal, weights, dbanks=dbs, dws=dws, dss=dss)
            nws = int(_hip.lib.hs_s2w_train_workspace(b, fh, fw, arr, len(meta))) if any(need_w) else 0
            ws = torch.empty(nws, device=signal.device, dtype=torch.uint8) if nws else None
            st = _hip.lib.hs_s2w_train_bwd(signal.data_ptr(), b, c, fh, fw, arr, len(meta), dsig.data_ptr() if need_s else None,
                                           ws.data_ptr() if ws is not None else None, nws, _hip.stream_ptr())
            _hip.check(st, 'hs_s2w_train_bwd')
        return (None, dsig) + tuple(dws)


_CHECK_LABELS = os.environ.get('HS_CHECK_LABELS', '0') == '1'


class PixelCrossEntropy(torch.autograd.Function):
    """F.cross_entropy(logits, target, ignore_index=..., reduction='none') for fp32 / bf16 (N, C, H, W) CUDA logits, one launch per
    direction (hs_cross_entropy_typed_fwd / _bwd; stock: log-softmax + gather and their adjoints, after a widening cast for bf16).
    The loss is fp32 and the arithmetic f32 for either storage type; the logits' gradient has the logits' type.
    ``target`` must be int64 of shape (N, H, W) -- checked here, as F.cross_entropy does: the kernel walks N*H*W pixels of BOTH
    tensors.  A label outside [0, C) other than ``ignore_index`` contributes a zero loss and a zero gradient (the stock CUDA kernel
    fires a device-side assert for it); ``HS_CHECK_LABELS=1`` makes this function verify the range first (one host read, debugging)."""

    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        if logits.dim() < 2 or tuple(target.shape) != (logits.shape[0],) + tuple(logits.shape[2:]):
            raise ValueError(f'Expected target of shape {(logits.shape[0],) + tuple(logits.shape[2:])} for logits of shape '
                             f'{tuple(logits.shape)}, got {tuple(target.shape)}')
        if target.dtype != torch.int64:
            raise ValueError(f'PixelCrossEntropy: expected int64 class indices, got {target.dtype}')
        if target.device != logits.device:
            raise ValueError(f'PixelCrossEntropy: target on {target.device}, logits on {logits.device}')
        logits, target = logits.contiguous(), target.contiguous()
        n, c = logits.shape[:2]
        px = logits.numel() // (n * c)
        if _CHECK_LABELS and target.numel():
            bad = (target != int(ignore_index)) & ((target < 0) | (target >= c))
            if bool(bad.any()):
                raise IndexError(f'Target {int(target[bad][0])} is out of bounds for {c} classes (ignore_index {int(ignore_index)})')
        with _hip.device_scope(logits.device):
            loss = torch.empty(target.shape, device=logits.device, dtype=torch.float32)
            st = _hip.lib.hs_cross_entropy_typed_fwd(DTYPE_CODES[logits.dtype], logits.data_ptr(), target.data_ptr(), n, c, px, int(ignore_index),
                                                     loss.data_ptr(), _hip.stream_ptr())
            _hip.check(st, 'hs_cross_entropy_typed_fwd')
        ctx.save_for_backward(logits, target)
        ctx.ignore_index = int(ignore_index)
        return loss

    @staticmethod
    def backward(ctx, g):
        logits, target = ctx.saved_tensors
        n, c = logits.shape[:2]
        px = logits.numel() // (n * c)
        g = g.contiguous().float()
        with _hip.device_scope(logits.device):
            dl = torch.empty_like(logits)
            st = _hip.lib.hs_cross_entropy_typed_bwd(DTYPE_CODES[logits.dtype], logits.data_ptr(), target.data_ptr(), n, c, px, ctx.ignore_index,
                                                     g.data_ptr(), dl.data_ptr(), _hip.stream_ptr())
            _hip.check(st, 'hs_cross_entropy_typed_bwd')
        return dl, None, None


class BootstrapMean(torch.autograd.Function):
    """The per-image reduction of the bootstrapped cross entropy (hyperseg/losses/bootstrapped_ce_loss.py:19-25) on the device with no
    sort and no host read (hs_bootstrap_mean_fwd / _bwd): capturable into a HIP graph as it is."""

    @staticmethod
    def forward(ctx, values, k, thresh):
        values = values.contiguous()
        n = values.numel()
        with _hip.device_scope(values.device):
            ws = torch.empty(int(_hip.lib.hs_bootstrap_mean_workspace()), device=values.device, dtype=torch.uint8)
            out = torch.empty(8, device=values.device, dtype=torch.float32)
            st = _hip.lib.hs_bootstrap_mean_fwd(values.data_ptr(), n, int(k), float(thresh), ws.data_ptr(), out.data_ptr(), _hip.stream_ptr())
            _hip.check(st, 'hs_bootstrap_mean_fwd')
        ctx.save_for_backward(values, out)
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        values, state = ctx.saved_tensors
        with _hip.device_scope(values.device):
            gv = torch.empty_like(values)
            st = _hip.lib.hs_bootstrap_mean_bwd(values.data_ptr(), values.numel(), state.data_ptr(),
                                                g.contiguous().float().data_ptr(), gv.data_ptr(), _hip.stream_ptr())
            _hip.check(st, 'hs_bootstrap_mean_bwd')
        return gv, None, None


class BootstrappedCrossEntropy(torch.autograd.Function):
    """``BootstrappedCrossEntropyLoss.forward`` (hyperseg/losses/bootstrapped_ce_loss.py:15-27) as ONE Function (round 6): logits (N, C, H, W)
    fp32 / bf16, target (N, H, W) int64 -> the batch mean of the per-image bootstrapped losses, a 0-dim fp32 tensor.  The same values as
    ``BootstrapMeanOfBatch.apply(PixelCrossEntropy.apply(...).flatten(1), k, thresh)`` bit for bit, with ONE launch backward
    (hs_bootstrapped_ce_bwd: the pixel weights are formed inside the cross entropy's adjoint) where the pair takes two and an (N, HW)
    gradient tensor between them.  (A forward that also took the selection's first histogram level while making the losses measured
    slower -- hs_train_aux.hip -- and is not in.)"""

    @staticmethod
    def forward(ctx, logits, target, ignore_index, k, thresh):
        if logits.dim() != 4 or tuple(target.shape) != (logits.shape[0],) + tuple(logits.shape[2:]):
            raise ValueError(f'Expected target of shape {(logits.shape[0],) + tuple(logits.shape[2:])} for logits of shape '
                             f'{tuple(logits.shape)}, got {tuple(target.shape)}')
        if target.dtype != torch.int64 or target.device != logits.device:
            raise ValueError(f'BootstrappedCrossEntropy: expected int64 class indices on {logits.device}, got {target.dtype} on {target.device}')
        logits, target = logits.contiguous(), target.contiguous()
        n, c = logits.shape[:2]
        px = logits.numel() // (n * c)
        if _CHECK_LABELS and target.numel():
            bad = (target != int(ignore_index)) & ((target < 0) | (target >= c))
            if bool(bad.any()):
                raise IndexError(f'Target {int(target[bad][0])} is out of bounds for {c} classes (ignore_index {int(ignore_index)})')
        with _hip.device_scope(logits.device):
            ws = torch.empty(n * int(_hip.lib.hs_bootstrap_mean_workspace()), device=logits.device, dtype=torch.uint8)
            loss = torch.empty(n, px, device=logits.device, dtype=torch.float32)
            out = torch.empty(n * 8 + 1, device=logits.device, dtype=torch.float32)              # (N, 8) state | the mean
            st = _hip.lib.hs_bootstrapped_ce_fwd(DTYPE_CODES[logits.dtype], logits.data_ptr(), target.data_ptr(), n, c, px, int(ignore_index),
                                                 int(k), float(thresh), ws.data_ptr(), loss.data_ptr(), out.data_ptr(),
                                                 out.data_ptr() + 4 * n * 8, _hip.stream_ptr())
            _hip.check(st, 'hs_bootstrapped_ce_fwd')
        ctx.save_for_backward(logits, target, loss, out)
        ctx.ignore_index = int(ignore_index)
        return out[n * 8:].view(())

    @staticmethod
    def backward(ctx, g):
        logits, target, loss, state = ctx.saved_tensors
        n, c = logits.shape[:2]
        px = logits.numel() // (n * c)
        if g.dtype != torch.float32 or not g.is_contiguous():
            g = g.float().contiguous()
        with _hip.device_scope(logits.device):
            dl = torch.empty_like(logits)
            st = _hip.lib.hs_bootstrapped_ce_bwd(DTYPE_CODES[logits.dtype], logits.data_ptr(), target.data_ptr(), n, c, px, ctx.ignore_index,
                                                 loss.data_ptr(), state.data_ptr(), g.data_ptr(), dl.data_ptr(), _hip.stream_ptr())
            _hip.check(st, 'hs_bootstrapped_ce_bwd')
        return dl, None, None, None, None


class MetaConvGeneral(torch.autograd.Function):
    """MetaConv2d with stride / dilation / any zero padding (hs_meta_conv_fwd) under autograd: backward = hs_meta_conv_bwd (input-
    and per-sample weight-gradient gather kernels).  fp32; the non-zero padding modes are padded by the caller with F.pad, whose
    adjoint autograd already has (the reference does exactly that, meta_conv.py:175-181)."""

    @staticmethod
    def forward(ctx, x, w, c_out, kernel_size, stride, pads, dilation, groups):
        # ``w`` may be MetaSequential's column range w[:, a:b] of a wider weight tensor: rows ``stride(0)`` apart, unit column stride --
        # the kernel takes the row stride (ldw), so such a view is read in place (it is NOT ``is_contiguous()``)
        x, w = x.contiguous().float(), (w if w.dim() == 2 and w.stride(1) == 1 else w.contiguous()).float()
        if not (w.is_cuda and w.device == x.device and w.dim() == 2 and w.stride(1) == 1):
            raise _hip.HipLibraryError(f'MetaConvGeneral: w must be a 2-D CUDA tensor on {x.device} with unit column stride, got '
                                       f'{tuple(w.shape)} strides {tuple(w.stride())} on {w.device}')
        (kh, kw), (sh, sw), (pt, pb, pl, pr), (dh, dw) = kernel_size, stride, pads, dilation
        b, cin, h, wd = x.shape
        ho = (h + pt + pb - dh * (kh - 1) - 1) // sh + 1
        wo = (wd + pl + pr - dw * (kw - 1) - 1) // sw + 1
        if ho <= 0 or wo <= 0:
            raise ValueError(f'kernel {kernel_size} (dilation {dilation}) does not fit the padded {h}x{wd} input')
        y = torch.empty(b, c_out, ho, wo, device=x.device, dtype=torch.float32)
        with _hip.device_scope(x.device):
            st = _hip.lib.hs_meta_conv_fwd(_hip.dev_ptr(x, 'x'), b, cin, h, wd, w.data_ptr(), w.stride(0), c_out, kh, kw,
                                           sh, sw, pt, pb, pl, pr, dh, dw, 0, groups, None, y.data_ptr(), _hip.stream_ptr())
            _hip.check(st, 'hs_meta_conv_fwd')
        ctx.save_for_backward(x, w)
        ctx.meta = (c_out, kernel_size, stride, pads, dilation, groups)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        c_out, (kh, kw), (sh, sw), (pt, pb, pl, pr), (dh, dw), groups = ctx.meta
        dy = dy.contiguous().float()
        b, cin, h, wd = x.shape
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        rows = c_out * (cin // groups) * kh * kw
        dwt = None
        if ctx.needs_input_grad[1]:
            dwt = (torch.empty if rows == w.shape[1] else torch.zeros)(w.shape, device=w.device, dtype=torch.float32)
        with _hip.device_scope(x.device):
            st = _hip.lib.hs_meta_conv_bwd(x.data_ptr(), b, cin, h, wd, w.data_ptr(), w.stride(0), c_out, kh, kw, sh, sw, pt, pb, pl, pr,
                                           dh, dw, groups, dy.data_ptr(), dx.data_ptr() if dx is not None else None,
                                           dwt.data_ptr() if dwt is not None else None, dwt.stride(0) if dwt is not None else 0,
                                           _hip.stream_ptr())
            _hip.check(st, 'hs_meta_conv_bwd')
        return dx, dwt, None, None, None, None, None, None


def meta_conv_general(x, w, c_out, kernel_size, stride, padding, dilation, padding_mode, groups):
    """The general MetaConv2d under autograd (meta_conv.py:163-186): explicit F.pad for the non-zero modes in the reference's own
    (quirky) order -- ``padding + padding`` = (ph, pw, ph, pw) read by F.pad as (left, right, top, bottom) -- then the zero-padding
    Function above."""
    import torch.nn.functional as F
    ph, pw = padding
    if padding_mode != 'zeros' and (ph or pw):
        x = F.pad(x, (ph, pw, ph, pw), mode=padding_mode)
        pads = (0, 0, 0, 0)
    else:
        pads = (ph, ph, pw, pw)
    return MetaConvGeneral.apply(x, w, c_out, tuple(kernel_size), tuple(stride), pads, tuple(dilation), groups)


class BootstrapMeanBatched(torch.autograd.Function):
    """BootstrapMean for every image of a batch in one set of launches (hs_bootstrap_mean_batched_fwd / _bwd: grid.y = image):
    values (N, n) -> (N,) per-image losses.  Nine launches per step whatever the batch (the per-image form: nine per image)."""

    @staticmethod
    def forward(ctx, values, k, thresh):
        values = values.contiguous()
        imgs, n = values.shape
        with _hip.device_scope(values.device):
            ws = torch.empty(imgs * int(_hip.lib.hs_bootstrap_mean_workspace()), device=values.device, dtype=torch.uint8)
            out = torch.empty(imgs, 8, device=values.device, dtype=torch.float32)
            st = _hip.lib.hs_bootstrap_mean_batched_fwd(values.data_ptr(), imgs, n, int(k), float(thresh), ws.data_ptr(), out.data_ptr(),
                                                        _hip.stream_ptr())
            _hip.check(st, 'hs_bootstrap_mean_batched_fwd')
        ctx.save_for_backward(values, out)
        return out[:, 0].clone()

    @staticmethod
    def backward(ctx, g):
        values, state = ctx.saved_tensors
        imgs, n = values.shape
        with _hip.device_scope(values.device):
            gv = torch.empty_like(values)
            st = _hip.lib.hs_bootstrap_mean_batched_bwd(values.data_ptr(), imgs, n, state.data_ptr(), g.contiguous().float().data_ptr(),
                                                        gv.data_ptr(), _hip.stream_ptr())
            _hip.check(st, 'hs_bootstrap_mean_batched_bwd')
        return gv, None, None


class BootstrapMeanOfBatch(torch.autograd.Function):
    """The loss module's batch form (round 5): values (N, n) -> the MEAN of the N per-image bootstrapped losses, a 0-dim tensor, from the
    same launches as BootstrapMeanBatched plus a one-thread mean; the adjoint takes the one upstream gradient as it is.  Replaces
    ``BootstrapMeanBatched.apply(...).sum() / N`` and what autograd makes of it: a clone, a reduction, two scalar multiplications, a
    contiguous copy of the expanded gradient -- six stock launches of a config-5 step."""

    @staticmethod
    def forward(ctx, values, k, thresh):
        values = values.contiguous()
        imgs, n = values.shape
        with _hip.device_scope(values.device):
            ws = torch.empty(imgs * int(_hip.lib.hs_bootstrap_mean_workspace()), device=values.device, dtype=torch.uint8)
            out = torch.empty(imgs * 8 + 1, device=values.device, dtype=torch.float32)          # (imgs, 8) state | the mean
            st = _hip.lib.hs_bootstrap_mean_of_batch_fwd(values.data_ptr(), imgs, n, int(k), float(thresh), ws.data_ptr(), out.data_ptr(),
                                                         out.data_ptr() + 4 * imgs * 8, _hip.stream_ptr())
            _hip.check(st, 'hs_bootstrap_mean_of_batch_fwd')
        ctx.save_for_backward(values, out)
        return out[imgs * 8:].view(())

    @staticmethod
    def backward(ctx, g):
        values, state = ctx.saved_tensors
        imgs, n = values.shape
        if g.dtype != torch.float32 or not g.is_contiguous():
            g = g.float().contiguous()
        with _hip.device_scope(values.device):
            gv = torch.empty_like(values)
            st = _hip.lib.hs_bootstrap_mean_of_batch_bwd(values.data_ptr(), imgs, n, state.data_ptr(), g.data_ptr(), gv.data_ptr(), _hip.stream_ptr())
            _hip.check(st, 'hs_bootstrap_mean_of_batch_bwd')
        return gv, None, None


def patch_conv_apply(*args):
    """``PatchConv.apply`` behind the autocast-dtype check."""
    if torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') != torch.bfloat16:
        # there is no fp16 storage type
        raise NotImplementedError(f"patch conv under torch.autocast('cuda', dtype={torch.get_autocast_dtype('cuda')}): only "
                                  'torch.bfloat16 autocast is supported (fp32 without autocast)')
    return PatchConv.apply(*args)


class UpsampleBilinear(torch.autograd.Function):
    """F.interpolate(x, size, mode='bilinear', align_corners=False) for fp32 / bf16 CUDA tensors, forward and adjoint on the HIP
    kernels (hs_upsample_bilinear_fwd / _bf16_fwd / _typed_bwd): the decoder's final logits in training.  bf16 storage in, bf16
    storage out (f32 arithmetic, one rounding), as the stock op under autocast -- without its widen / narrow cast launches."""

    @staticmethod
    def forward(ctx, x, size):
        ctx.meta = (tuple(x.shape), tuple(size), x.dtype)
        x = x.contiguous()
        if x.dtype == torch.float32:
            return HF.upsample_bilinear(x, size)
        b, c, hi, wi = x.shape
        with _hip.device_scope(x.device):
            y = torch.empty(b, c, size[0], size[1], device=x.device, dtype=torch.bfloat16)
            st = _hip.lib.hs_upsample_bilinear_bf16_fwd(_hip.dev_ptr(x, 'x', torch.bfloat16), b, c, hi, wi, size[0], size[1], y.data_ptr(),
                                                        _hip.stream_ptr())
            _hip.check(st, 'hs_upsample_bilinear_bf16_fwd')
        return y

    @staticmethod
    def backward(ctx, dy):
        (b, c, hi, wi), (ho, wo), dt = ctx.meta
        dy = dy.contiguous().to(dt)
        with _hip.device_scope(dy.device):
            dx = torch.empty(b, c, hi, wi, device=dy.device, dtype=dt)
            st = _hip.lib.hs_upsample_bilinear_typed_bwd(DTYPE_CODES[dt], dy.data_ptr(), 0, b, c, hi, wi, ho, wo, dx.data_ptr(), _hip.stream_ptr())
            _hip.check(st, 'hs_upsample_bilinear_typed_bwd')
        return dx, None


def upsample_bilinear(x, size):
    """F.interpolate(x, size, mode='bilinear', align_corners=False), differentiable: own kernels for up-sampling CUDA tensors in
    fp32 / bf16 storage (the decoder's final logits upsample in training), the stock op otherwise."""
    import torch.nn.functional as F
    if (x.is_cuda and x.dtype in (torch.float32, torch.bfloat16) and size[0] >= x.shape[2] and size[1] >= x.shape[3]
            and x.shape[0] * x.shape[1] <= 65535):
        return UpsampleBilinear.apply(x, tuple(size))
    return F.interpolate(x, size, mode='bilinear', align_corners=False)


class BankSlices(torch.autograd.Function):
    """The three column ranges [0, r1), [r1, r2), [r2, r3) of an inverted residual's bank (pw1 | depthwise | pw3) as views.  Plain
    slicing gives the same views, but autograd then returns each range's gradient through its own zeros + copy and sums the three
    (8 launches per level and step); here the backward is ONE concatenation."""

    @staticmethod
    def forward(ctx, bank, r1, r2, r3):
        ctx.meta = (tuple(bank.shape), r1, r2, r3)
        outs = bank[:, :r1], bank[:, r1:r2], bank[:, r2:r3]
        ctx.grads = None
        if USE_SHARED_BANK_GRAD and bank.is_cuda:
            ctx.grads = _BankGradBuffer(bank.shape, ((0, r1), (r1, r2), (r2, r3)))
            for i, o in enumerate(outs):
                o._hs_grad_slot = (ctx.grads, i)           # read by the consuming Function's forward (the object the caller passes on)
        return outs

    @staticmethod
    def backward(ctx, g1, g2, g3):
        (rows, cols), r1, r2, r3 = ctx.meta
        if ctx.grads is not None:
            buf, owned = ctx.grads.buf, ctx.grads.owns((g1, g2, g3))
            ctx.grads.buf = None                            # a second backward pass fills a new one
            ctx.grads.taken = set()
            if owned:
                return buf, None, None, None
        ref = next(g for g in (g1, g2, g3) if g is not None)
        parts = [g if g is not None else ref.new_zeros(rows, w) for g, w in ((g1, r1), (g2, r2 - r1), (g3, r3 - r2))]
        if cols > r3:
            parts.append(ref.new_zeros(rows, cols - r3))
        return torch.cat(parts, dim=1), None, None, None


class BankPack(torch.autograd.Function):
    """(B, hp_total, fh, fw) reference-layout weights -> patch-major bank (B*fh*fw, ld); backward is the transpose.
    The re-layout runs in fp32 (a bf16 weight tensor produced under autocast is widened first) and the bank stays fp32 into
    PatchConv, so neither adds a rounding of its own."""

    @staticmethod
    def forward(ctx, w, rows):
        ctx.shape = tuple(w.shape)
        ctx.rows = rows
        ctx.dtype = w.dtype
        return HF.bank_pack(w.float().contiguous() if w.dtype != torch.float32 else w, 0, rows)

    @staticmethod
    def backward(ctx, dbank):
        b, c, fh, fw = ctx.shape
        if dbank.is_cuda and dbank.dtype == torch.float32 and dbank.stride(1) == 1 and b <= 65535:
            with _hip.device_scope(dbank.device):                  # one LDS-tiled transpose incl. the zero tail (hs_bank_unpack_fwd)
                dw = torch.empty(ctx.shape, device=dbank.device, dtype=torch.float32)
                st = _hip.lib.hs_bank_unpack_fwd(dbank.data_ptr(), dbank.stride(0), b, c, fh, fw, 0, ctx.rows, dw.data_ptr(), _hip.stream_ptr())
                _hip.check(st, 'hs_bank_unpack_fwd')
            return dw.to(ctx.dtype), None
        dw = dbank.new_zeros(ctx.shape, dtype=ctx.dtype)
        dw[:, :ctx.rows] = dbank[:, :ctx.rows].reshape(b, fh, fw, ctx.rows).permute(0, 3, 1, 2).to(ctx.dtype)
        return dw, None


def patch_conv_train(x, weight, c_out, k, padding, padding_mode, groups, hp):
    """Differentiable MetaPatchConv2d core: ``weight`` is the reference-layout tensor (B, >=hp, fh, fw)."""
    fh, fw = weight.shape[-2:]
    bank = BankPack.apply(weight, hp)
    return patch_conv_apply(x, bank, (fh, fw), c_out, k, padding, padding_mode, groups)


class StageMaterialize(torch.autograd.Function):
    """cat(coords, skip, bilinear(prev)) as ONE launch under autograd (hs_stage_input_typed_fwd, the materialising twin of the
    kernels' prologue): the stock formulation is two linspaces, a stack, the interpolation and a concatenation per level and
    step -- five to six launches.  Backward: the skip's gradient is a channel range of dy (a view), the previous level's is the
    bilinear adjoint (hs_upsample_bilinear_typed_bwd) of its range; coordinates are constants.  Under bf16 autocast (or with a
    bf16 previous level) the previous level is read as stored and the result is written as bf16 -- what the first convolution
    reads -- and the adjoint runs bf16 -> bf16: no cast launch on either side (18 of them per config-5 step before)."""

    @staticmethod
    def forward(ctx, skip, prev, coords):
        low = (torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16) or \
            (prev is not None and prev.dtype == torch.bfloat16) or skip.dtype == torch.bfloat16
        stage = HF.StageInput(skip.contiguous().float(), prev.contiguous() if prev is not None else None, coords=coords)
        ctx.meta = (tuple(skip.shape), tuple(prev.shape) if prev is not None else None, bool(coords), skip.dtype,
                    prev.dtype if prev is not None else None)
        return stage.materialize(torch.bfloat16 if low else torch.float32)

    @staticmethod
    def backward(ctx, dy):
        (b, cs, h, w), pshape, coords, sdt, pdt = ctx.meta
        off = 2 if coords else 0
        dskip = dy[:, off:off + cs].to(sdt) if ctx.needs_input_grad[0] else None
        dprev = None
        if pshape is not None and ctx.needs_input_grad[1]:
            _, cp, hp, wp = pshape
            if (hp, wp) == (h, w):
                dprev = dy[:, off + cs:].to(pdt)
            else:
                # the previous level's channel range of dy is read IN PLACE (batch stride = all channels): no slice copy
                dyc = dy if dy.is_contiguous() else dy.contiguous()
                if dyc.dtype not in DTYPE_CODES:
                    dyc = dyc.float()
                with _hip.device_scope(dy.device):
                    dprev = torch.empty(b, cp, hp, wp, device=dy.device, dtype=dyc.dtype)
                    st = _hip.lib.hs_upsample_bilinear_typed_bwd(DTYPE_CODES[dyc.dtype], dyc.data_ptr() + dyc.element_size() * (off + cs) * h * w,
                                                                 dyc.shape[1] * h * w, b, cp, hp, wp, h, w, dprev.data_ptr(), _hip.stream_ptr())
                    _hip.check(st, 'hs_upsample_bilinear_typed_bwd')
                dprev = dprev.to(pdt)
        return dskip, dprev, None


def materialize_stage(stage):
    """cat(coords, skip, bilinear(prev)), differentiable (training only): one HIP launch where the shapes allow (CUDA, an up-sampling
    or same-size previous level), stock ops otherwise."""
    import torch.nn.functional as F
    skip, prev = stage.skip, stage.prev
    b, _, h, w = skip.shape
    ok = skip.is_cuda and skip.dtype in (torch.float32, torch.bfloat16) and b * max(skip.shape[1], 1) <= 65535
    if ok and prev is not None:
        ok = prev.dtype in (torch.float32, torch.bfloat16) and h >= prev.shape[2] and w >= prev.shape[3] and \
            prev.shape[0] * prev.shape[1] <= 65535
    if ok and USE_HIP_STAGE:
        return StageMaterialize.apply(skip, prev, stage.coords)
    parts = []
    if stage.coords:
        cx = torch.linspace(-1, 1, steps=w, device=skip.device)
        cy = torch.linspace(-1, 1, steps=h, device=skip.device)
        parts.append(torch.stack([cx.view(1, w).expand(h, w), cy.view(h, 1).expand(h, w)], 0).unsqueeze(0).expand(b, -1, -1, -1))
    parts.append(skip)
    if prev is not None:
        if prev.shape[-2:] != skip.shape[-2:]:
            if (prev.is_cuda and prev.dtype == torch.float32 and h >= prev.shape[2] and w >= prev.shape[3]
                    and prev.shape[0] * prev.shape[1] <= 65535):
                prev = UpsampleBilinear.apply(prev, (h, w))
            else:
                prev = F.interpolate(prev, (h, w), mode='bilinear', align_corners=False)
        parts.append(prev)
    return torch.cat(parts, dim=1)


USE_HIP_STAGE = True        # tests switch it off to compare with the stock formulation
USE_HIP_S2W_TRAIN = True    # the decoder's banks for training from one launch (S2WBanksTrain); off: one grouped conv (GEMM) per level
