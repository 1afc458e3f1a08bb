"""Autograd nodes of the DPT dense decoder on the HIP engine.

Maps are channels-last ([B, H, W, C], the token-major layout of the ViT carried to the upsampled
resolutions) in the engine's operand type.  Every convolution is a GEMM of libssl4gie_hip.so:
  * Conv2d(k=1)            -> engine.LinearFn on [B*H*W, Cin]
  * ConvTranspose2d(k = s) -> GEMM against W[(i, j, co), ci] + pixel-shuffle scatter
  * Conv2d(k=3, pad=1)     -> GEMM over the (dy, dx, ci) patch matrix (ssl4gie_im2col3x3); the data
                              gradient is the same product against the flipped kernel (stride 1)
                              or the patch-matrix transpose (stride 2); the weight gradient is the
                              split-K TN product dY^T cols with the bias gradient fused in.
The patch matrix is recomputed in backward instead of being kept (it is 9x the activation).
Weight re-layouts (a few MB, once per optimizer step) and the scatter of dW back into the
parameter's [Cout, Cin, 3, 3] layout are host-side tensor plumbing; all activation-sized work is
HIP.  Reference: Models/DPT_decoder.py (lines cited per node).
"""
from __future__ import annotations

import contextlib
import os
import weakref

import torch

from . import ops
from .engine import GradSink, LPCache, touched_since, weights_epoch, wgrad_fork


# SSL4GIE_IMPLICIT_CONV=0 forces the materialised patch matrix (A/B measurements, parity tests)
_IMPLICIT = os.environ.get("SSL4GIE_IMPLICIT_CONV", "1") != "0"
# SSL4GIE_DIRECT_CONV=0 sends the narrow head convolutions through the GEMM paths again (A/B)
_DIRECT = os.environ.get("SSL4GIE_DIRECT_CONV", "1") != "0"


def _fills_chip(x, stride, n_out):
    """the gathered NT kernel only exists with 256x256 tiles: below ~100 tiles (ResNet layer4's
    7x7 maps) the 128x128 kernel over a (small) materialised patch matrix keeps more CUs busy"""
    B, H, W, _ = x.shape
    Ho, Wo = ops.conv_out_hw(H, W, stride)
    return ((B * Ho * Wo + 255) // 256) * ((n_out + 255) // 256) >= 96


def _w_direct(lp, weight, dtype):
    """[Cout, 9 Cin] operand of the direct kernels (taps row-major, channels innermost, unpadded)"""
    Cout = weight.shape[0]
    return _derived(lp, weight, "c3x", dtype, lambda w: ops.conv3x3_weight_pack(w, dtype, 0), recipe=(0, None))


def _derived(lp: LPCache, p: torch.Tensor, tag: str, dtype, fn, recipe=None):
    """operand-type tensor derived from parameter `p` by `fn`, cached until p changes.
    recipe = (mode, ld) of ops.conv3x3_weight_pack when `fn` is exactly that: the image is then also refreshed,
    together with every other stale one, by ONE launch at the top of the next forward (refresh_conv_operands)."""
    key = (id(p), tag)
    ver = (p._version, p.data_ptr(), dtype, weights_epoch())
    if recipe is not None:
        rec = lp.__dict__.setdefault("_conv_recipes", {})
        if key not in rec:
            rec[key] = (weakref.ref(p), recipe[0], recipe[1], dtype)
    ent = lp._c.get(key)
    if ent is None or ent[0] != ver:
        with torch.no_grad():
            t = fn(p.detach())
            if t.dtype != dtype:
                t = ops.cast(t.contiguous(), dtype)
            else:
                t = t.contiguous()
        ent = (ver, t)
        lp._c[key] = ent
    return ent[1]


def refresh_conv_operands(lp: LPCache):
    """Re-pack every stale 3x3-convolution operand image registered with _derived(recipe=...) in ONE launch
    (ssl4gie_conv3x3_weight_pack_batch) — a ResNet-50 otherwise issues ~60 small pack kernels per optimizer step,
    each in front of its convolution.  Runs when the fused optimizers have moved the weights (the engine's
    weights epoch); images made stale any other way are re-packed lazily, one by one, as before."""
    rec = lp.__dict__.get("_conv_recipes")
    if not rec or not _BATCH_PACK:
        return
    epoch = weights_epoch()
    seen = lp.__dict__.get("_conv_epoch")
    if seen == epoch:
        return
    lp.__dict__["_conv_epoch"] = epoch
    moved = touched_since(seen) if seen is not None else None   # None: any parameter may have moved
    items, dead = [], []
    for key, (ref, mode, ld, dtype) in rec.items():
        p = ref()
        if p is None:
            dead.append(key)
            continue
        ver = (p._version, p.data_ptr(), dtype, epoch)
        ent = lp._c.get(key)
        if ent is not None and ent[0] == ver:
            continue
        if ent is not None and moved is not None and key[0] not in moved and ent[0][:3] == ver[:3]:
            lp._c[key] = (ver, ent[1])   # only the epoch moved, and not for this parameter: the image stands
            continue
        if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or p.dim() != 4:
            continue
        items.append((key, ver, p, mode, ld, dtype))
    for key in dead:
        rec.pop(key, None)
        lp._c.pop(key, None)
    for dtype in {it[5] for it in items}:
        group = [it for it in items if it[5] == dtype]
        outs = ops.conv3x3_weight_pack_batch([it[2].detach() for it in group], dtype, [it[3] for it in group],
                                             [it[4] for it in group])
        for it, t in zip(group, outs):
            lp._c[it[0]] = (it[1], t)


# SSL4GIE_CONV_PACK_BATCH=0: every 3x3 operand image by its own launch again (A/B)
_BATCH_PACK = os.environ.get("SSL4GIE_CONV_PACK_BATCH", "1") != "0"


def _pad_cols(m: torch.Tensor, ld: int) -> torch.Tensor:
    if m.shape[1] == ld:
        return m
    out = m.new_zeros(m.shape[0], ld)
    out[:, :m.shape[1]] = m
    return out


def _write_grad(target: torch.Tensor, value: torch.Tensor, accumulate: bool):
    if accumulate:
        target.add_(value)
    else:
        target.copy_(value)


class TokensToMapFn(torch.autograd.Function):
    """Slice(1) -> Transpose -> Unflatten (DPT_decoder.py:5-11, :328-332, :449-459): fp32 tap
    [B, 1+L, D] -> [B*L, D] operand rows (= the [B, 14, 14, D] map)."""

    @staticmethod
    def forward(ctx, z, dtype):
        ctx.shape = z.shape
        return ops.tokens_to_map(z.contiguous(), dtype)

    @staticmethod
    def backward(ctx, dx):
        B, L1, D = ctx.shape
        return ops.map_to_tokens(dx.contiguous(), B, L1 - 1, D), None


class Conv3x3Fn(torch.autograd.Function):
    """y = conv3x3(act(x)) + bias, pad 1, stride 1 or 2; act = ReLU if relu_in
    (layer*_rn :412-447, ResidualConvUnit_custom :212-233, output_conv.0/.2 :469-478,
    act_postprocess42.1 :397-403)."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, relu_in, sink: GradSink, lp: LPCache, want_stats=False):
        """want_stats: also return BatchNorm partial statistics of y (see engine.LinearFn) or None"""
        B, H, W, Cin = x.shape
        Cout = weight.shape[0]
        dt = x.dtype
        ld = ops.k_pad(9 * Cin, dt)
        w2 = _derived(lp, weight, f"c3:{ld}", dt, lambda w: ops.conv3x3_weight_pack(w, dt, 0, ld), recipe=(0, ld))
        x = x.contiguous()
        b = bias.detach() if bias is not None else None
        ctx.save_for_backward(x, weight, bias)
        ctx.cfg = (stride, relu_in, sink, lp, ld)
        implicit = _IMPLICIT and ld == 9 * Cin and ops.conv3x3_implicit_ok(x, stride, Cout)
        Ho, Wo = ops.conv_out_hw(H, W, stride)
        stats = None
        # direct kernel: where it is ahead of the gathered GEMM (ops.conv3x3_direct_ok), and instead
        # of a materialised patch matrix where the gathered GEMM does not apply (layer1_rn: Cin = 96)
        direct = _DIRECT and stride == 1 and (ops.conv3x3_direct_ok(x, Cout) or
                                              (not implicit and ops.conv3x3_direct_supported(x, Cout)))
        if direct and want_stats and b is None:
            # narrow ResNet layers (64 -> 64 @56, 128 -> 128 @28): direct kernel, statistics per tile
            y, stats = ops.conv3x3_direct_fwd(x, w2 if ld == 9 * Cin else _w_direct(lp, weight, dt), None,
                                              relu_in, colstats=True)
            ctx.mark_non_differentiable(stats)
            ctx.set_materialize_grads(False)  # no zero tensor for the statistics' (absent) gradient
        elif want_stats and b is None and dt == torch.bfloat16 and Cout % 8 == 0 and ld % 64 == 0:
            # the statistics only exist in the 256x256 kernel: taken whatever the tile count
            if implicit:
                y, stats = ops.conv3x3_fwd(x, w2, None, stride, relu_in, colstats=True)
            else:
                y, stats = ops.linear_fwd(ops.im2col3x3(x, stride, relu_in, ld), w2, None, out_dtype=dt,
                                          colstats=True)
                y = y.view(B, Ho, Wo, Cout)
            ctx.mark_non_differentiable(stats)
            ctx.set_materialize_grads(False)  # no zero tensor for the statistics' (absent) gradient
        elif direct:
            # narrow layer (output_conv.2: 128 -> 32): halo-in-LDS kernel instead of a GEMM tile that
            # would be 7/8 padding
            y = ops.conv3x3_direct_fwd(x, w2 if ld == 9 * Cin else _w_direct(lp, weight, dt), b, relu_in)
        elif implicit and _fills_chip(x, stride, Cout):
            y = ops.conv3x3_fwd(x, w2, b, stride, relu_in)  # patch matrix gathered in the GEMM
        else:
            y = ops.linear_fwd(ops.im2col3x3(x, stride, relu_in, ld), w2, b, out_dtype=dt).view(B, Ho, Wo, Cout)
        return (y, stats) if want_stats else y

    @staticmethod
    def backward(ctx, dy, *unused):
        if dy is None:  # the map itself was not used downstream (gradients are not materialised)
            return (None,) * 8
        x, weight, bias = ctx.saved_tensors
        stride, relu_in, sink, lp, ld = ctx.cfg
        B, H, W, Cin = x.shape
        Cout = weight.shape[0]
        dt = x.dtype
        dy = dy.contiguous()
        dy2 = dy.view(-1, Cout)
        (tw, tb), acc, rets = sink.plan([weight, bias])
        side = wgrad_fork(sink, (weight, bias), dy, x, weight, bias) if tw is not None else None
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            Conv3x3Fn._wgrad(tw, tb, acc, dy2, x, stride, relu_in, ld, Cin, Cout)
        dx = None
        if ctx.needs_input_grad[0]:
            return Conv3x3Fn._dgrad(ctx, dy, dy2, x, weight, stride, relu_in, lp, ld, B, H, W, Cin, Cout, dt, rets)
        return dx, rets[0], rets[1], None, None, None, None, None

    @staticmethod
    def _wgrad(tw, tb, acc, dy2, x, stride, relu_in, ld, Cin, Cout):
        if tw is not None:
            fuse_b = tb is not None and not acc  # the bias gradient rides on the same product
            if _DIRECT and stride == 1 and ops.conv3x3_direct_wgrad_ok(x, Cout):
                dw2 = ops.conv3x3_direct_wgrad(dy2, x, relu_in, bias_out=tb if fuse_b else None)
            elif _IMPLICIT and ops.conv3x3_implicit_ok(x, stride, Cout, wgrad=True):
                dw2 = ops.conv3x3_bwd_weight(dy2, x, stride, relu_in, bias_out=tb if fuse_b else None)
            else:
                cols = ops.im2col3x3(x, stride, relu_in, ld)  # recomputed, not kept
                dw2 = ops.linear_bwd_weight(dy2, cols, bias_out=tb if fuse_b else None)
                del cols
            if tb is not None and not fuse_b:
                ops.colsum(dy2, out=tb, accumulate=True)
            if dw2.dtype == torch.float32 and dw2.stride(1) == 1 and tw.is_contiguous():
                ops.conv3x3_wgrad_unpack(dw2, tw, acc)
            else:
                _write_grad(tw, dw2[:, :9 * Cin].view(Cout, 3, 3, Cin).permute(0, 3, 1, 2), acc)
        elif tb is not None:
            ops.colsum(dy2, out=tb, accumulate=acc)

    @staticmethod
    def _dgrad(ctx, dy, dy2, x, weight, stride, relu_in, lp, ld, B, H, W, Cin, Cout, dt, rets):
        if True:
            if stride == 1:
                ld2 = ops.k_pad(9 * Cout, dt)
                wd = _derived(lp, weight, f"c3d:{ld2}", dt, lambda w: ops.conv3x3_weight_pack(w, dt, 1, ld2),
                              recipe=(1, ld2))
                dy4 = dy.view(B, H, W, Cout)
                if _DIRECT and ops.conv3x3_direct_ok(dy4, Cin):
                    # the data gradient is the same direct kernel on dy with the flipped weight
                    wdd = _derived(lp, weight, "c3dd", dt, lambda w: ops.conv3x3_weight_pack(w, dt, 1),
                                   recipe=(1, None))
                    dxr = ops.conv3x3_direct_fwd(dy4, wdd, None, relu_mask=x if relu_in else None)
                    return dxr, rets[0], rets[1], None, None, None, None, None
                if _IMPLICIT and ld2 == 9 * Cout and _fills_chip(dy4, 1, Cin) and \
                        ops.conv3x3_implicit_ok(dy4, 1, Cin):
                    # the ReLU in front of this convolution masks its data gradient in the epilogue
                    dxr = ops.conv3x3_fwd(dy4, wd, None, 1, False, relu_mask=x if relu_in else None)
                    return dxr, rets[0], rets[1], None, None, None, None, None
                else:
                    dcols = ops.im2col3x3(dy4, 1, False, ld2)
                    dxr = ops.linear_fwd(dcols, wd, None, out_dtype=dt).view(B, H, W, Cin)
            else:
                w2 = _derived(lp, weight, f"c3:{ld}", dt, lambda w: ops.conv3x3_weight_pack(w, dt, 0, ld), recipe=(0, ld))
                w2t = _derived(lp, weight, f"c3t:{ld}", dt, lambda w: ops.conv3x3_weight_pack(w, dt, 2, ld),
                               recipe=(2, ld))
                dcols = ops.linear_bwd_data(dy2, w2, w2t)  # [M_out, ld]
                dxr = ops.col2im3x3(dcols, B, H, W, Cin, stride)
            dx = ops.relu_bwd(x, dxr) if relu_in else dxr
        return dx, rets[0], rets[1], None, None, None, None, None


class ConvTransposeFn(torch.autograd.Function):
    """ConvTranspose2d(kernel = stride = k) on a [B*H*W, Cin] map (act_postprocess12.1 :345-354,
    act_postprocess22.1 :371-380): GEMM against W2[(i, j, co), ci] = weight[ci, co, i, j], then a
    pixel-shuffle scatter that also adds the bias."""

    @staticmethod
    def forward(ctx, x2, weight, bias, B, H, W, sink: GradSink, lp: LPCache):
        Cin, Cout, k, _ = weight.shape
        dt = x2.dtype
        ld = ops.k_pad(Cin, dt)
        if ld != Cin:
            # Cin = 96 (act_postprocess1) is not a whole number of the bf16 GEMM's 64-deep K-tiles: with
            # zero-padded operand copies (6 MB) the product runs on the MFMA bf16 kernel instead of the
            # generic fp32 one (164 -> ~20 us)
            w2 = _derived(lp, weight, f"ct:{ld}", dt,
                          lambda w: _pad_cols(w.permute(2, 3, 1, 0).reshape(k * k * Cout, Cin), ld))
            g = ops.linear_fwd(_pad_cols(x2, ld), w2, None, out_dtype=dt)
        else:
            w2 = _derived(lp, weight, "ct", dt, lambda w: w.permute(2, 3, 1, 0).reshape(k * k * Cout, Cin))
            g = ops.linear_fwd(x2, w2, None, out_dtype=dt)
        y = ops.pixel_shuffle(g, bias.detach(), B, H, W, k, Cout)
        ctx.save_for_backward(x2, weight, bias)
        ctx.cfg = (B, H, W, sink, lp)
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, weight, bias = ctx.saved_tensors
        B, H, W, sink, lp = ctx.cfg
        Cin, Cout, k, _ = weight.shape
        dt = x2.dtype
        dy = dy.contiguous()
        dg = ops.pixel_unshuffle(dy, k)  # [B*H*W, k*k*Cout]
        (tw, tb), acc, rets = sink.plan([weight, bias])
        if tw is not None:
            dw2 = ops.linear_bwd_weight(dg, x2)  # [k*k*Cout, Cin]
            _write_grad(tw, dw2.view(k, k, Cout, Cin).permute(3, 2, 0, 1), acc)
        if tb is not None:
            ops.colsum(dy.view(-1, Cout), out=tb, accumulate=acc)
        dx = None
        if ctx.needs_input_grad[0]:
            w2 = _derived(lp, weight, "ct", dt, lambda w: w.permute(2, 3, 1, 0).reshape(k * k * Cout, Cin))
            w2t = _derived(lp, weight, "ct_t", dt,
                           lambda w: w.permute(2, 3, 1, 0).reshape(k * k * Cout, Cin).t())
            dx = ops.linear_bwd_data(dg, w2, w2t)
        return dx, rets[0], rets[1], None, None, None, None, None


class Upsample2xFn(torch.autograd.Function):
    """bilinear x2, align_corners=True (FeatureFusionBlock_custom.forward :293-295; Interpolate)."""

    @staticmethod
    def forward(ctx, x):
        return ops.bilinear2x_fwd(x.contiguous())

    @staticmethod
    def backward(ctx, dy):
        return ops.bilinear2x_bwd(dy.contiguous())


class ForkFn(torch.autograd.Function):
    """x -> (x, x) for a tensor with two consumers (the input of a residual unit feeds its first convolution AND
    its skip add, DPT_decoder.py:212-233): autograd would otherwise sum the two incoming gradients itself, with a
    torch clone + add kernel per fork (7 per depth step); here the sum is the library's element-wise kernel."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None or gb is None:
            return ga if gb is None else gb
        return ops.eltwise_add(ga.contiguous(), gb.contiguous())


class AddFn(torch.autograd.Function):
    """skip_add.add (:233, :290): FloatFunctional in float mode is a plain add."""

    @staticmethod
    def forward(ctx, a, b):
        return ops.eltwise_add(a.contiguous(), b.contiguous())

    @staticmethod
    def backward(ctx, g):
        return g, g


class DepthHeadFn(torch.autograd.Function):
    """ReLU -> Conv2d(32, 1, 1) -> Sigmoid (output_conv.3-.5, :479-481) -> fp32 [B, 1, H, W]."""

    @staticmethod
    def forward(ctx, x, weight, bias, sink: GradSink):
        B, H, W, C = x.shape
        x2 = x.contiguous().view(-1, C)
        w = weight.detach().reshape(-1).contiguous()
        y = ops.depth_head_fwd(x2, w, bias.detach())
        ctx.save_for_backward(x2, weight, bias, y)
        ctx.cfg = (B, H, W, sink)
        return y.view(B, 1, H, W)

    @staticmethod
    def backward(ctx, dy):
        x2, weight, bias, y = ctx.saved_tensors
        B, H, W, sink = ctx.cfg
        (tw, tb), acc, rets = sink.plan([weight, bias])
        dw = tw.view(-1) if tw is not None else torch.empty(weight.numel(), device=x2.device)
        db = tb if tb is not None else torch.empty(1, device=x2.device)
        dx = ops.depth_head_bwd(x2, weight.detach().reshape(-1).contiguous(), y,
                                dy.contiguous().view(-1).float(), dw, db, acc)
        return dx.view(B, H, W, -1), rets[0], rets[1], None


class SegHeadFn(torch.autograd.Function):
    """Conv2d(C, num_classes, 1) -> bilinear x2 (output_conv.4-.5 of the seg head, :483-497) ->
    fp32 logits [B, num_classes, 2H, 2W].  num_classes is tiny (1 for the binary task): the 1x1
    convolution runs as a GEMM against the weight padded to 8 output channels so that the map stays
    16-byte aligned through the bilinear kernel; the padding channels are dropped at the end."""

    @staticmethod
    def forward(ctx, x, weight, bias, sink: GradSink, lp: LPCache):
        B, H, W, C = x.shape
        nc = weight.shape[0]
        ncp = (nc + 7) // 8 * 8
        dt = x.dtype
        x2 = x.contiguous().view(-1, C)

        def pad_w(w):
            out = w.new_zeros(ncp, C)
            out[:nc] = w.reshape(nc, C)
            return out
        w8 = _derived(lp, weight, f"seg:{ncp}", dt, pad_w)
        b8 = torch.zeros(ncp, dtype=torch.float32, device=x.device)
        b8[:nc] = bias.detach()
        y = ops.linear_fwd(x2, w8, b8, out_dtype=dt)
        up = ops.bilinear2x_fwd(y.view(B, H, W, ncp))
        ctx.save_for_backward(x2, weight, bias)
        ctx.cfg = (B, H, W, nc, ncp, sink, lp)
        return up[..., :nc].permute(0, 3, 1, 2).float().contiguous()

    @staticmethod
    def backward(ctx, dy):
        x2, weight, bias = ctx.saved_tensors
        B, H, W, nc, ncp, sink, lp = ctx.cfg
        C = x2.shape[1]
        dt = x2.dtype
        g = torch.zeros(B, 2 * H, 2 * W, ncp, dtype=dt, device=dy.device)
        g[..., :nc] = dy.permute(0, 2, 3, 1)
        dy8 = ops.bilinear2x_bwd(g).view(-1, ncp)
        (tw, tb), acc, rets = sink.plan([weight, bias])
        if tw is not None:
            db8 = torch.empty(ncp, dtype=torch.float32, device=dy.device)
            dw8 = ops.linear_bwd_weight(dy8, x2, bias_out=db8)
            _write_grad(tw, dw8[:nc].view_as(tw), acc)
            if tb is not None:
                _write_grad(tb, db8[:nc], acc)
        dx = None
        if ctx.needs_input_grad[0]:
            def pad_w(w):
                out = w.new_zeros(ncp, C)
                out[:nc] = w.reshape(nc, C)
                return out
            w8 = _derived(lp, weight, f"seg:{ncp}", dt, pad_w)
            dx = ops.linear_bwd_data(dy8, w8).view(B, H, W, C)
        return dx, rets[0], rets[1], None, None
