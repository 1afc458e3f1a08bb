"""Loss functions of the finetune heads.  BASELINE.json's north_star leaves loss reductions host-side
("Host code stays Python on PyTorch-ROCm for the DataLoader, optimizer step and loss reductions"):
the torch formulations below are that path, and the parity reference.  On the HIP device the two
module classes run fused device kernels instead (csrc/loss_ops.hip — value and
gradient in five / three launches instead of ~100 small ones); SSL4GIE_FUSED_LOSS=0 or CPU tensors
select the torch formulation.  Same call signature and numerics as the reference's
`Depth_estimation/Metrics/losses.py` (scale-and-shift-invariant depth loss, :120-146): per-image
closed-form 2x2 least squares for (scale, shift) on the valid pixels (:5-25), masked MSE / (2M)
(:51-57) and alpha x a 4-scale masked gradient L1 (:60-77, :104-117), batch-based reduction
(:28-38).  Written without the reference's data-dependent `nonzero()` indexing, so a step has no
device -> host synchronisation; results are identical (images with a singular system get
scale = shift = 0, an empty mask contributes 0).
"""
from __future__ import annotations

import torch
import torch.nn as nn


def compute_scale_and_shift(prediction, target, mask):
    mask = mask.to(prediction.dtype)
    a_00 = torch.sum(mask * prediction * prediction, (1, 2))
    a_01 = torch.sum(mask * prediction, (1, 2))
    a_11 = torch.sum(mask, (1, 2))
    b_0 = torch.sum(mask * prediction * target, (1, 2))
    b_1 = torch.sum(mask * target, (1, 2))
    det = a_00 * a_11 - a_01 * a_01
    ok = det != 0
    safe = torch.where(ok, det, torch.ones_like(det))
    x_0 = torch.where(ok, (a_11 * b_0 - a_01 * b_1) / safe, torch.zeros_like(det))
    x_1 = torch.where(ok, (-a_01 * b_0 + a_00 * b_1) / safe, torch.zeros_like(det))
    return x_0, x_1


def _batch_based(image_loss, M):
    divisor = torch.sum(M)
    total = torch.sum(image_loss)
    return torch.where(divisor == 0, torch.zeros_like(total), total / torch.clamp(divisor, min=1e-30))


def mse_loss(prediction, target, mask):
    mask = mask.to(prediction.dtype)
    M = torch.sum(mask, (1, 2))
    res = prediction - target
    return _batch_based(torch.sum(mask * res * res, (1, 2)), 2 * M)


def gradient_loss(prediction, target, mask):
    mask = mask.to(prediction.dtype)
    M = torch.sum(mask, (1, 2))
    diff = mask * (prediction - target)
    grad_x = torch.abs(diff[:, :, 1:] - diff[:, :, :-1]) * (mask[:, :, 1:] * mask[:, :, :-1])
    grad_y = torch.abs(diff[:, 1:, :] - diff[:, :-1, :]) * (mask[:, 1:, :] * mask[:, :-1, :])
    return _batch_based(torch.sum(grad_x, (1, 2)) + torch.sum(grad_y, (1, 2)), M)


def _fused_ok(*ts):
    import os
    return all(t.is_cuda for t in ts) and os.environ.get("SSL4GIE_FUSED_LOSS", "1") != "0"


class _SsiFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, prediction, target, alpha, scales):
        import ctypes  # noqa: F401
        from . import _lib
        from .ops import ptr, stream
        L = _lib.load()
        p = prediction.contiguous().float()
        t = target.contiguous().float()
        B, H, W = p.shape
        nb = L.ssl4gie_ssi_loss_workspace_bytes(B, H, W)
        ws = torch.empty(nb, dtype=torch.uint8, device=p.device)
        loss = torch.empty((), dtype=torch.float32, device=p.device)
        dpred = torch.empty_like(p)
        _lib.check(L.ssl4gie_ssi_loss(ptr(p), ptr(t), ptr(loss), ptr(dpred), B, H, W, float(alpha), int(scales),
                                      ptr(ws), stream()), "ssi_loss")
        ctx.save_for_backward(dpred)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return dpred * g, None, None, None


class _DiceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, smooth):
        from . import _lib
        from .ops import ptr, stream
        L = _lib.load()
        B = logits.shape[0]
        l = logits.contiguous().float().view(B, -1)
        t = targets.contiguous().float().view(B, -1)
        ws = torch.empty(L.ssl4gie_dice_loss_workspace_bytes(B), dtype=torch.uint8, device=l.device)
        loss = torch.empty((), dtype=torch.float32, device=l.device)
        dl = torch.empty_like(l)
        _lib.check(L.ssl4gie_dice_loss(ptr(l), ptr(t), ptr(loss), ptr(dl), B, l.shape[1], float(smooth), ptr(ws),
                                       stream()), "dice_loss")
        ctx.save_for_backward(dl)
        ctx.shape = logits.shape
        return loss

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return (dl * g).view(ctx.shape), None, None


class ScaleAndShiftInvariantLoss(nn.Module):
    def __init__(self, alpha=0.5, scales=4, reduction="batch-based"):
        super().__init__()
        if reduction != "batch-based":
            raise NotImplementedError("the reference trains with the batch-based reduction only")
        self.alpha = alpha
        self.scales = scales
        self.prediction_ssi = None

    def forward(self, prediction, target):
        prediction = prediction.squeeze(1).float()
        target = target.squeeze(1).float()
        if _fused_ok(prediction, target) and 1 <= self.scales <= 4:
            self.prediction_ssi = None  # not materialised by the fused kernels
            return _SsiFn.apply(prediction, target, self.alpha, self.scales)
        mask = target > 0
        scale, shift = compute_scale_and_shift(prediction, target, mask)
        ssi = scale.view(-1, 1, 1) * prediction + shift.view(-1, 1, 1)
        self.prediction_ssi = ssi
        total = mse_loss(ssi, target, mask)
        if self.alpha > 0:
            reg = 0
            for s in range(self.scales):
                step = 2 ** s
                reg = reg + gradient_loss(ssi[:, ::step, ::step], target[:, ::step, ::step],
                                          mask[:, ::step, ::step])
            total = total + self.alpha * reg
        return total


class SoftDiceLoss(torch.nn.Module):
    """Mirror of `Binary_segmentation/Metrics/losses.py:5-24` (host-side loss reduction): soft Dice
    on sigmoid(logits), mean over the batch."""

    def __init__(self, smooth=1e-8):
        super().__init__()
        self.smooth = smooth

    def forward(self, logits, targets):
        if _fused_ok(logits, targets):
            return _DiceFn.apply(logits, targets, self.smooth)
        num = targets.size(0)
        m1 = torch.sigmoid(logits).view(num, -1)
        m2 = targets.view(num, -1)
        score = 2.0 * ((m1 * m2).sum(1) + self.smooth) / ((m1 * m1).sum(1) + (m2 * m2).sum(1) + self.smooth)
        return 1 - score.sum() / num
