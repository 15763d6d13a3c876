"""Training path of the dynamic patch-wise convolution: ``torch.autograd.Function`` wrappers around the HIP forward
and backward kernels (SURVEY.md section 8b "Autograd", Appendix E; BASELINE config 5).

The reference has no custom backward: autograd differentiates its ATen ops (F.pad, unfold, grouped conv2d, fold).
Here the convolution itself -- forward, per-patch weight gradient, input gradient -- runs in HIP kernels; the cheap glue
around it in TRAINING mode (stage-input concatenation, BatchNorm with batch statistics, activations, the grouped 1x1
``signal2weights`` convolution) stays stock PyTorch so that autograd composes the whole decoder.  Inference never comes
through this module: it uses the fused kernels (one launch per level).
"""
import ctypes as C

import torch

from . import _hip
from . import functional as HF


def needs_grad(*tensors):
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


class PatchConv(torch.autograd.Function):
    """y = patch_conv(x, bank): Op A / Op B with plain tensors.  Saves x and the bank; backward launches
    hs_patch_conv_bwd_input / hs_patch_conv_bwd_weight."""

    @staticmethod
    def forward(ctx, x, bank, grid, c_out, k, padding, padding_mode, groups):
        x = x.contiguous()
        if bank.stride(1) != 1:
            bank = bank.contiguous()
        y = HF.patch_conv(x, grid, bank, c_out, k, padding, padding_mode, groups)
        ctx.save_for_backward(x, bank)
        ctx.meta = (tuple(grid), c_out, k, padding, padding_mode, groups)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, bank = ctx.saved_tensors
        (fh, fw), c_out, k, pad, mode, groups = ctx.meta
        dy = dy.contiguous()
        b, c_in, h, w = x.shape
        stream = _hip.stream_ptr()
        dx = dbank = None
        bank_ptr, ld = HF._bank_ptr(bank)
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            st = _hip.lib.hs_patch_conv_bwd_input(_hip.dev_ptr(dy, 'dy'), bank_ptr, ld, b, c_in, h, w, fh, fw, c_out, k,
                                                  pad, HF.PAD_MODES[mode], groups, dx.data_ptr(), stream)
            _hip.check(st, 'hs_patch_conv_bwd_input')
        if ctx.needs_input_grad[1]:
            rows = c_out * (c_in // groups) * k * k
            full = torch.zeros(bank.shape[0], bank.shape[1], device=x.device, dtype=torch.float32)
            st = _hip.lib.hs_patch_conv_bwd_weight(_hip.dev_ptr(x, 'x'), _hip.dev_ptr(dy, 'dy'), b, c_in, h, w, fh, fw,
                                                   c_out, k, pad, HF.PAD_MODES[mode], groups, full.data_ptr(),
                                                   full.stride(0), stream)
            _hip.check(st, 'hs_patch_conv_bwd_weight')
            dbank = full
            assert rows <= full.shape[1]
        return dx, dbank, None, None, None, None, None, None


class BankPack(torch.autograd.Function):
    """(B, hp_total, fh, fw) reference-layout weights -> patch-major bank (B*fh*fw, ld); backward is the transpose."""

    @staticmethod
    def forward(ctx, w, rows):
        ctx.shape = tuple(w.shape)
        ctx.rows = rows
        return HF.bank_pack(w, 0, rows)

    @staticmethod
    def backward(ctx, dbank):
        b, c, fh, fw = ctx.shape
        dw = dbank.new_zeros(ctx.shape)
        dw[:, :ctx.rows] = dbank[:, :ctx.rows].reshape(b, fh, fw, ctx.rows).permute(0, 3, 1, 2)
        return dw, None


def patch_conv_train(x, weight, c_out, k, padding, padding_mode, groups, hp):
    """Differentiable MetaPatchConv2d core: ``weight`` is the reference-layout tensor (B, >=hp, fh, fw)."""
    fh, fw = weight.shape[-2:]
    bank = BankPack.apply(weight, hp)
    return PatchConv.apply(x, bank, (fh, fw), c_out, k, padding, padding_mode, groups)


def materialize_stage(stage):
    """cat(coords, skip, bilinear(prev)) with stock differentiable ops (training only)."""
    import torch.nn.functional as F
    skip, prev = stage.skip, stage.prev
    b, _, h, w = skip.shape
    parts = []
    if stage.coords:
        cx = torch.linspace(-1, 1, steps=w, device=skip.device)
        cy = torch.linspace(-1, 1, steps=h, device=skip.device)
        parts.append(torch.stack([cx.view(1, w).expand(h, w), cy.view(h, 1).expand(h, w)], 0).unsqueeze(0).expand(b, -1, -1, -1))
    parts.append(skip)
    if prev is not None:
        if prev.shape[-2:] != skip.shape[-2:]:
            prev = F.interpolate(prev, (h, w), mode='bilinear', align_corners=False)
        parts.append(prev)
    return torch.cat(parts, dim=1)
