"""Autograd nodes of the detection backbone's feature pyramid on the HIP engine (SURVEY §8f rank 1;
reference `ViTDet_FPN`, Models/models.py:213-259).

Maps are channels-last [B, H, W, C] in the engine's operand type.  Convolutions reuse the DPT nodes
(1x1 = LinearFn, 3x3 = Conv3x3Fn implicit GEMM, ConvTranspose2d(2, 2) = ConvTransposeFn); this file
adds what is particular to the pyramid: MaxPool2d(2), GELU on a map and nn.LayerNorm((C, H, W)) —
a per-image normalisation over the whole map whose weight and bias are full [C, H, W] tensors.  The
LayerNorm kernels want the affine parameters in the map's own element order, so an [H, W, C] fp32
copy of the [C, H, W] parameters is cached (refreshed when the parameter changes) and the parameter
gradients are permuted back when they are written to the arena.
"""
from __future__ import annotations

import torch

from . import ops
from .dpt_engine import _derived, _write_grad
from .engine import GradSink, LPCache


class MaxPool2Fn(torch.autograd.Function):
    """nn.MaxPool2d(2) (models.py:218)"""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return ops.maxpool2x2_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.maxpool2x2_bwd(x, dy.contiguous())


class GeluMapFn(torch.autograd.Function):
    """nn.GELU on a map (models.py:241), exact erf form"""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return ops.gelu_map(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.gelu_map(x, dy.contiguous())


def _hwc(lp, p):
    """[H, W, C] fp32 view of a (C, H, W) table: the parameter's own memory when it is stored channels-last
    (ViTDet_FPN), else a permuted copy cached until the parameter changes"""
    v = p.detach().permute(1, 2, 0)
    if v.is_contiguous() and v.dtype == torch.float32:
        return v
    return _derived(lp, p, "mapln", torch.float32, lambda q: q.permute(1, 2, 0))


class MapLayerNormFn(torch.autograd.Function):
    """nn.LayerNorm((C, H, W)) on a channels-last map: weight / bias are [C, H, W] parameters."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, sink: GradSink, lp: LPCache):
        x = x.contiguous()
        w, b = _hwc(lp, weight), _hwc(lp, bias)
        y, mean, rstd = ops.map_layernorm_fwd(x, w, b, eps)
        ctx.save_for_backward(x, weight, bias, mean, rstd)
        ctx.cfg = (sink, lp)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias, mean, rstd = ctx.saved_tensors
        sink, lp = ctx.cfg
        w = _hwc(lp, weight)
        (tw, tb), acc, rets = sink.plan([weight, bias])
        C, H, W = weight.shape

        # the kernel writes (or accumulates into) the gradient where it belongs when that memory is channels-last
        # already (ViTDet_FPN's tables in the arena): no [H, W, C] temporary, no permuted copy back
        views = [t.permute(1, 2, 0) for t in (tw, tb) if t is not None]
        direct = bool(views) and all(v.is_contiguous() for v in views)
        if direct:
            dw = tw.permute(1, 2, 0) if tw is not None else None
            db = tb.permute(1, 2, 0) if tb is not None else None
        else:
            dw = torch.empty(H, W, C, dtype=torch.float32, device=x.device) if tw is not None else None
            db = torch.empty(H, W, C, dtype=torch.float32, device=x.device) if tb is not None else None
        dx = ops.map_layernorm_bwd(x, dy.contiguous(), w, mean, rstd, dw, db, accumulate=direct and acc)
        if not direct:
            if tw is not None:
                _write_grad(tw, dw.permute(2, 0, 1), acc)
            if tb is not None:
                _write_grad(tb, db.permute(2, 0, 1), acc)
        return dx, rets[0], rets[1], None, None, None
