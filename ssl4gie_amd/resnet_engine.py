"""Autograd nodes of the ResNet50 trunk on the HIP engine.

Channels-last maps in the engine's operand type.  conv1 (7x7 s2) is a GEMM over the stem patch
matrix, 1x1 convolutions are token-major GEMMs (engine.LinearFn, after a stride-2 row pick for the
downsample branches), 3x3 convolutions reuse dpt_engine.Conv3x3Fn; BatchNorm2d runs in training
mode with fp32 batch statistics and fuses ReLU and the bottleneck's residual add.
Reference: torchvision 0.10 resnet.py (un-vendored) as instantiated at
Models/models.py:63-75 and Models/moco_v3/main_moco.py:185-187.
"""
from __future__ import annotations

import weakref

import torch

from . import ops
from .dpt_engine import _derived, _pad_cols, _write_grad
from .engine import touched_since, weights_epoch, GradSink, LPCache


class _StemCols:
    """The stem operands (bf16: the packed image, 0.1 GB; fp32: the patch matrix) of the last two image batches (1.2 GB each at 256 x 224 x 224 — small
    change against 288 GB of HBM).  MoCo feeds the same two views to the base and the momentum
    encoder (moco/builder.py:127-135) and the weight gradient needs the matrix again: one im2col
    per view and step instead of three (the entry is dropped once its weight gradient is taken).  Entries are tied to the image tensor OBJECT (weak
    reference + version counter), never to its address, so a recycled allocation cannot hit."""

    def __init__(self, keep=2):
        self.keep, self.items = keep, []

    def get(self, imgs, dtype, make=None):
        """`make` (default: the patch matrix) builds the cached value; entries are keyed on it too"""
        make = make or ops.stem_im2col7x7
        key = (dtype, getattr(make, "__name__", "make"))
        for i, (ref, ver, dt, val) in enumerate(self.items):
            if ref() is imgs and ver == imgs._version and dt == key:
                self.items.append(self.items.pop(i))
                return val
        val = make(imgs, dtype) if make is ops.stem_im2col7x7 else make(imgs)
        self.items.append((weakref.ref(imgs), imgs._version, key, val))
        del self.items[:-self.keep]
        return val

    def drop(self, imgs):
        """forget this batch's matrix: its weight gradient has been taken, the training step that
        produced it is over — a later forward on the SAME tensor object (a benchmark loop on a fixed
        synthetic batch) must pay for its own im2col, as a step on fresh images does"""
        self.items = [it for it in self.items if it[0]() is not imgs]

    def clear(self):
        self.items.clear()


STEM_COLS = _StemCols()
# SSL4GIE_STEM_DIRECT=0: the bf16 stem through the patch matrix again (A/B)
_STEM_DIRECT = __import__("os").environ.get("SSL4GIE_STEM_DIRECT", "1") != "0"


class StemConvFn(torch.autograd.Function):
    """conv1: Conv2d(3, 64, 7, stride 2, pad 3, bias=False) on the fp32 NCHW image."""

    @staticmethod
    def forward(ctx, imgs, weight, dtype, sink: GradSink, lp: LPCache, want_stats=False):
        """want_stats: also return the BatchNorm partial statistics of the map (or None)"""
        imgs = imgs.contiguous().float()
        B = imgs.shape[0]
        Cout = weight.shape[0]
        if _STEM_DIRECT and dtype == torch.bfloat16 and Cout == 64 and imgs.shape[1] == 3:
            # bf16: no patch matrix — the image is packed once per batch (bf16, 4 channels, padded) and
            # read by the direct stem kernels (csrc/conv_direct.hip): 43 + 179 us instead of 740 + 454
            H, W = imgs.shape[2:]
            packed = STEM_COLS.get(imgs, dtype, ops.stem7x7_pack)
            w2s = _derived(lp, weight, "stem7", dtype, ops.stem7x7_weight)
            ctx.save_for_backward(imgs, weight)
            ctx.cfg = (dtype, sink)
            ctx.direct = True
            if want_stats:
                y, stats = ops.stem7x7_fwd(packed, w2s, B, H, W, colstats=True)
                ctx.mark_non_differentiable(stats)
                ctx.set_materialize_grads(False)
                return y, stats
            return ops.stem7x7_fwd(packed, w2s, B, H, W)
        ctx.direct = False
        cols, Ho, Wo = STEM_COLS.get(imgs, dtype)
        ld = cols.shape[1]
        w2 = _derived(lp, weight, f"stem:{ld}", dtype,
                      lambda w: _pad_cols(w.permute(0, 2, 3, 1).reshape(Cout, 147), ld))
        ctx.save_for_backward(imgs, weight)
        ctx.cfg = (dtype, sink)
        stats = None
        if want_stats and ops.colstats_ok(cols.shape[0], Cout, ld, dtype):
            y, stats = ops.linear_fwd(cols, w2, None, out_dtype=dtype, colstats=True)
            ctx.mark_non_differentiable(stats)
            ctx.set_materialize_grads(False)  # no zero tensor for the statistics' (absent) gradient
        else:
            y = ops.linear_fwd(cols, w2, None, out_dtype=dtype)
        y = y.view(B, Ho, Wo, Cout)
        return (y, stats) if want_stats else y

    @staticmethod
    def backward(ctx, dy, *unused):
        if dy is None:  # the map itself was not used downstream (gradients are not materialised)
            return (None,) * 6
        imgs, weight = ctx.saved_tensors
        dtype, sink = ctx.cfg
        Cout = weight.shape[0]
        (tw,), acc, rets = sink.plan([weight])
        if tw is not None and ctx.direct:
            B, _, H, W = imgs.shape
            packed = STEM_COLS.get(imgs, dtype, ops.stem7x7_pack)  # cached for the last two batches
            _write_grad(tw, ops.stem7x7_wgrad(dy.contiguous(), packed, B, H, W), acc)
        elif tw is not None:
            cols, _, _ = STEM_COLS.get(imgs, dtype)  # cached for the last two batches, else recomputed
            dw2 = ops.linear_bwd_weight(dy.contiguous().view(-1, Cout), cols)
            _write_grad(tw, dw2[:, :147].view(Cout, 7, 7, 3).permute(0, 3, 1, 2), acc)
        STEM_COLS.drop(imgs)
        return None, rets[0], None, None, None, None


class Subsample2Fn(torch.autograd.Function):
    """the pixel pick of a stride-2 1x1 convolution (Bottleneck.downsample[0]).  join: engine.GradJoin of
    the block input, which also feeds conv1: the scattered gradient is deposited there and summed in
    conv1's data-gradient GEMM epilogue (EPI_ADD_AUX) instead of by an autograd add kernel over the map."""

    @staticmethod
    def forward(ctx, x, join=None):
        ctx.hw = x.shape[1:3]
        ctx.join = join if (join is not None and join.claim() == "depositor") else None
        return ops.subsample2(x.contiguous())

    @staticmethod
    def backward(ctx, dy):
        g = ops.subsample2_bwd(dy.contiguous(), *ctx.hw)
        if ctx.join is not None:
            ctx.join.deposit(g)
            return None, None
        return g, None


SYNC_BN_COLLECTIVES = [0]  # torch.distributed collectives issued by SyncBatchNorm layers (rehearsal / tests)
SYNC_BN_DIRECT = [0]       # exchanges carried by the library's peer-to-peer all-gather instead (SSL4GIE_SYNCBN=direct)


def combine_batch_stats(mean_l, var_l, count_l, group=None):
    """Exchange step of SyncBatchNorm: per-rank (mean, biased var, count) -> global (mean, biased var,
    count) by the pooled-variance identity (ranks may hold different row counts).  One all_gather of
    2C + 1 floats into one [W, 2C+1] buffer; everything, the total count included,
    stays on the device (`total` is a 0-d tensor: no host synchronisation per layer).  Pure torch on
    [W, 2C+1] floats; runs on gloo for the CPU tests."""
    import torch.distributed as dist
    C = mean_l.numel()
    world = dist.get_world_size(group)
    packed = torch.empty(2 * C + 1, dtype=mean_l.dtype, device=mean_l.device)
    packed[:C] = mean_l
    packed[C:2 * C] = var_l
    packed[2 * C] = float(count_l)
    flat = torch.empty(world * (2 * C + 1), dtype=mean_l.dtype, device=mean_l.device)
    dist.all_gather_into_tensor(flat, packed, group=group)
    SYNC_BN_COLLECTIVES[0] += 1
    g = flat.view(world, 2 * C + 1)
    n = g[:, 2 * C:]           # [W, 1]
    total = n.sum()
    wgt = n / total            # [W, 1]
    mean = (g[:, :C] * wgt).sum(0)
    var = ((g[:, C:2 * C] + (g[:, :C] - mean) ** 2) * wgt).sum(0)
    return mean, var, total


def sync_batch_stats(mean_l, var_l, rows, bn, group=None):
    """SyncBatchNorm's forward exchange -> (mean, rstd, total rows as a 0-d device tensor), running
    statistics updated.  On the GPU the pooled combine, rstd and the running statistics are ONE kernel
    (ssl4gie_bn_combine_stats) behind the all-gather of the (mean, var, count) records — torch.distributed's
    all_gather_into_tensor (RCCL) by default, the library's peer-to-peer all-gather with SSL4GIE_SYNCBN=direct /
    auto; round 6: the torch.distributed path no longer runs its ~20 element-wise torch kernels per layer.
    CPU tensors (gloo tests) take the torch formulation of the same arithmetic."""
    from . import parallel
    mom = bn.momentum if bn.momentum is not None else 0.1
    if not mean_l.is_cuda:
        mean, var, total = combine_batch_stats(mean_l, var_l, rows, group)
        rstd = torch.rsqrt(var + bn.eps)
        if bn.running_mean is not None:
            with torch.no_grad():  # unbiased variance: n / (n - 1), on the device
                bn.running_mean.mul_(1 - mom).add_(mean, alpha=mom)
                bn.running_var.mul_(1 - mom).add_(var * (total / (total - 1.0).clamp_(min=1.0)), alpha=mom)
        return mean, rstd, total
    import torch.distributed as dist
    from . import _lib
    ex = parallel.syncbn_exchange(group)
    world = ex.world if ex is not None else dist.get_world_size(group)
    C = mean_l.numel()
    rec = torch.empty(2 * C + 1, dtype=torch.float32, device=mean_l.device)
    rec[:C] = mean_l
    rec[C:2 * C] = var_l
    rec[2 * C] = float(rows)
    gathered = torch.empty(world * (2 * C + 1), dtype=torch.float32, device=mean_l.device)
    if ex is None:
        dist.all_gather_into_tensor(gathered, rec, group=group)
        SYNC_BN_COLLECTIVES[0] += 1
    else:
        ex.all_gather_(rec, gathered)
        SYNC_BN_DIRECT[0] += 1
    out = torch.empty(2 * C + 1, dtype=torch.float32, device=mean_l.device)
    mean, rstd, total = out[:C], out[C:2 * C], out[2 * C]
    rm = bn.running_mean if bn.running_mean is not None else None
    _lib.check(_lib.load().ssl4gie_bn_combine_stats(
        gathered.data_ptr(), world, C, float(bn.eps), float(mom), ops.ptr(rm),
        ops.ptr(bn.running_var if rm is not None else None), mean.data_ptr(), rstd.data_ptr(), total.data_ptr(),
        ops.stream()), "bn_combine_stats")
    return mean, rstd, total


def sync_sum(sums, group=None):
    """SyncBatchNorm's backward exchange: element-wise sum over ranks of the [2, C] local sums"""
    import torch.distributed as dist
    from . import parallel
    ex = parallel.syncbn_exchange(group) if sums.is_cuda else None
    if ex is None:
        g = sums.clone()
        dist.all_reduce(g, group=group)
        SYNC_BN_COLLECTIVES[0] += 1
        return g
    flat = sums.contiguous().view(-1)
    gathered = torch.empty(ex.world * flat.numel(), dtype=torch.float32, device=sums.device)
    ex.all_gather_(flat, gathered)
    SYNC_BN_DIRECT[0] += 1
    return gathered.view(ex.world, -1).sum(0).view_as(sums)   # rank order: bitwise the same on every rank


def _sync_group(bn):
    import torch.distributed as dist
    if isinstance(bn, torch.nn.SyncBatchNorm) and dist.is_available() and dist.is_initialized() \
            and dist.get_world_size(bn.process_group) > 1:
        return True, bn.process_group
    return False, None


# nn.BatchNorm's `num_batches_tracked` only matters for momentum=None (cumulative average) and in
# checkpoints.  A `+= 1` on the device buffer per layer and forward is 212 tiny kernels per MoCo-R50
# step (1.1 ms): the count is kept on the host instead and written to the buffer when somebody can
# see it — state_dict() (a pre-hook), or at once for momentum=None layers.  Code that reads the
# buffer directly between steps calls flush_batch_counts(model) first.
_PENDING_BATCHES = weakref.WeakKeyDictionary()


def _flush_one(bn, *_):
    n = _PENDING_BATCHES.pop(bn, 0)
    if n and bn.num_batches_tracked is not None:
        bn.num_batches_tracked += n


def _count_batch(bn):
    if bn.momentum is None:
        bn.num_batches_tracked += 1
        return
    if bn not in _PENDING_BATCHES:
        if not getattr(bn, "_ssl4gie_count_hook", False):
            bn.register_state_dict_pre_hook(_flush_one)
            bn._ssl4gie_count_hook = True
        _PENDING_BATCHES[bn] = 0
    _PENDING_BATCHES[bn] += 1


def flush_batch_counts(model):
    """write the host-side batch counts into every BatchNorm's `num_batches_tracked` buffer"""
    for m in model.modules():
        if m in _PENDING_BATCHES:
            _flush_one(m)


# SSL4GIE_SYNCBN_FUSED=0: under SyncBatchNorm (world_size > 1) the bit-map backward, the stem's bn -> relu -> maxpool
# pass and the momentum encoder's statistics-only + affine-epilogue products fall back to the separate passes
# (rounds 4-5 behaviour: A/B of what BASELINE config 3 costs in its multi-rank form)
_SYNC_FUSED = __import__("os").environ.get("SSL4GIE_SYNCBN_FUSED", "1") != "0"
# SSL4GIE_BN_BITS=0: BatchNorm + residual + ReLU backward reads the ReLU output for its mask again (A/B)
_BN_BITS = __import__("os").environ.get("SSL4GIE_BN_BITS", "1") != "0"
# SSL4GIE_BN_XMASK=0: read the ReLU output for the mask of residual-free BatchNorm + ReLU layers again (A/B)
_XMASK = __import__("os").environ.get("SSL4GIE_BN_XMASK", "1") != "0"


def _bn_params_moved(epoch, gamma, beta):
    """whether gamma / beta were rewritten since `epoch` (engine.touched_since: MoCo's EMA update between the base
    encoder's forward and its backward moves the MOMENTUM encoder only and does not count; an optimizer step does)"""
    if weights_epoch() == epoch:
        return False
    moved = touched_since(epoch)
    if moved is None:
        return True
    return any(p is not None and id(p) in moved for p in (gamma, beta))


class BatchNormFn(torch.autograd.Function):
    """nn.BatchNorm2d / BatchNorm1d / SyncBatchNorm over the rows of a [..., C] tensor, (+ residual)
    (+ ReLU).  With an nn.SyncBatchNorm holder and world_size > 1 the statistics are global: local
    statistics kernel -> all_gather + pooled combine -> apply kernel; backward: local reduction
    kernel -> all_reduce of the 2C sums -> apply kernel (dgamma / dbeta stay local, DDP averages)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, res, bn, relu, sink: GradSink, stats=None, join=None):
        """stats: per-128-row partial sums [parts, 2, C] of x from the GEMM that produced it
        (LinearFn / Conv3x3Fn with want_stats) — saves the statistics pass over x.
        join: engine.GradJoin of the residual input (its gradient is deposited, not returned)."""
        ctx.join = join if (join is not None and res is not None and join.claim() == "depositor") else None
        shp = x.shape
        C = shp[-1]
        x2 = x.contiguous().view(-1, C)
        r2 = res.contiguous().view(-1, C) if res is not None else None
        training = bn.training or bn.running_mean is None
        g = gamma.detach() if gamma is not None else None
        b = beta.detach() if beta is not None else None
        sync, group = _sync_group(bn)
        total = None  # SyncBatchNorm: 0-d device tensor, rows over all ranks
        bits = None
        if training and not sync and _BN_BITS and relu and r2 is not None and stats is not None \
                and x2.dtype == torch.bfloat16 and any(ctx.needs_input_grad):
            # bn3 of a bottleneck: the backward gets its ReLU mask as a bit map (1/16 of the output's bytes)
            mom = bn.momentum if bn.momentum is not None else 0.1
            y, bits, mean, rstd = ops.bn_fwd_bits(x2, g, b, r2, bn.running_mean, bn.running_var, mom, bn.eps, stats)
        elif training and not sync:
            mom = bn.momentum if bn.momentum is not None else 0.1
            y, mean, rstd = ops.bn_fwd(x2, g, b, r2, bn.running_mean, bn.running_var, mom, bn.eps, relu,
                                       True, partials=stats)
        elif training:
            mean_l, var_l = ops.bn_stats(x2, partials=stats)
            mean, rstd, total = sync_batch_stats(mean_l, var_l, x2.shape[0], bn, group)
            if _BN_BITS and _SYNC_FUSED and relu and r2 is not None and x2.dtype == torch.bfloat16 \
                    and any(ctx.needs_input_grad):
                # bn3 of a bottleneck under SyncBatchNorm: the same bit-map form as on one rank, fed with the
                # GLOBAL statistics the exchange returned (round 6)
                y, bits = ops.bn_apply_bits(x2, ops.bn_coef_stats(mean, rstd, g, b), r2)
            else:
                y, _, _ = ops.bn_fwd(x2, g, b, r2, None, None, 0.0, bn.eps, relu, False, mean, rstd)
        else:
            mean = bn.running_mean
            rstd = torch.rsqrt(bn.running_var + bn.eps)  # [C] floats: host-side plumbing
            y, _, _ = ops.bn_fwd(x2, g, b, r2, None, None, 0.0, bn.eps, relu, False, mean, rstd)
        if training and bn.num_batches_tracked is not None:
            _count_batch(bn)
        # the mask-from-x backward rebuilds the ReLU mask from gamma / beta as they are AT BACKWARD TIME: it is only
        # valid while no optimizer has stepped since this forward — checked there against this stamp;
        # that path never reads the ReLU output, so it is not saved for it either
        xmask = relu and res is None and _XMASK and bits is None
        ctx.wepoch = weights_epoch()
        ctx.save_for_backward(x2, bits if bits is not None else (y if (relu and not xmask) else None), gamma, beta,
                              mean, rstd)
        ctx.cfg = (shp, relu, res is not None, sink, training, sync, group, total)
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        x2, y, gamma, beta, mean, rstd = ctx.saved_tensors
        shp, relu, has_res, sink, training, sync, group, total = ctx.cfg
        if not training:
            raise NotImplementedError("backward through BatchNorm in eval mode (frozen statistics) is "
                                      "not built: the reference freezes trunks under no_grad")
        C = shp[-1]
        (tg, tb), acc, rets = sink.plan([gamma, beta])
        dy2 = dy.contiguous().view(-1, C)
        gd = gamma.detach() if gamma is not None else None
        if not sync and relu and has_res and y is not None and y.dtype == torch.uint8:
            dx, dres = ops.bn_bwd_bits(dy2, y, x2, gd, mean, rstd, tg, tb, acc)   # `y` holds the forward's bit map
        elif not sync and relu and not has_res and _XMASK:
            # the ReLU mask from x and the forward's coefficients: the ReLU output is not read again
            if _bn_params_moved(ctx.wepoch, gamma, beta):
                raise RuntimeError("BatchNorm parameters were updated between this forward and its backward: the "
                                   "ReLU mask rebuilt from them would disagree with the forward (set SSL4GIE_BN_XMASK=0)")
            dx, dres = ops.bn_bwd_xmask(dy2, x2, gd, beta.detach() if beta is not None else None, mean, rstd,
                                        tg, tb, acc), None
        elif not sync:
            dx, dres = ops.bn_bwd(dy2, y, x2, gd, mean, rstd, relu, has_res, tg, tb, acc)
        else:
            import torch.distributed as dist
            xm = relu and not has_res and _XMASK   # mask rebuilt from x and the (global) forward coefficients
            if xm and _bn_params_moved(ctx.wepoch, gamma, beta):
                raise RuntimeError("BatchNorm parameters were updated between this forward and its backward "
                                   "(set SSL4GIE_BN_XMASK=0)")
            bd = beta.detach() if beta is not None else None
            if xm:
                sums, dres = ops.bn_bwd_reduce_xmask(dy2, x2, gd, bd, mean, rstd), None
            elif relu and has_res and y is not None and y.dtype == torch.uint8:   # `y` holds the forward's bit map
                sums, dres = ops.bn_bwd_reduce_bits(dy2, y, x2, mean, rstd)
            else:
                sums, dres = ops.bn_bwd_reduce(dy2, y, x2, mean, rstd, relu, has_res)
            for tgt, row in ((tb, 0), (tg, 1)):  # parameter gradients are the LOCAL sums
                if tgt is not None:
                    if acc:
                        tgt.add_(sums[row])
                    else:
                        tgt.copy_(sums[row])
            gsums = sync_sum(sums, group)
            gsums /= total  # the 1 / count of the dx formula, folded into the sums on the device
            if xm:
                dx = ops.bn_bwd_apply_xmask(dy2, x2, gd, bd, mean, rstd, gsums, 1.0)
            elif relu and has_res:   # dres is the masked gradient: one tensor instead of dy and the ReLU output
                dx = ops.bn_bwd_apply(dres, None, x2, gd, mean, rstd, gsums, 1.0, False)
            else:
                dx = ops.bn_bwd_apply(dy2, y, x2, gd, mean, rstd, gsums, 1.0, relu)
        gres = dres.view(shp) if has_res else None
        if gres is not None and ctx.join is not None:
            ctx.join.deposit(gres)  # joined in the data-gradient GEMM of the block's first convolution
            gres = None
        return (dx.view(shp), rets[0], rets[1], gres, None, None, None, None, None)


class MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y, arg = ops.maxpool3x3s2_fwd(x.contiguous())
        ctx.save_for_backward(arg)
        ctx.hw = x.shape[1:3]
        return y

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        return ops.maxpool3x3s2_bwd(dy.contiguous(), arg, *ctx.hw)


class BnReluMaxPoolFn(torch.autograd.Function):
    """torchvision ResNet's bn1 -> relu -> maxpool(3, 2, 1) behind the stem convolution (`ResNet._forward_impl`)
    as ONE pass over the convolution's output: the training-mode statistics come from the convolution's epilogue
    partials, the normalisation + ReLU are applied inside the pool's window reads — the normalised map
    (0.4 GB per view at 256 x 112 x 112 x 64) is never written.  Backward: pool gradient by the saved argmax,
    BatchNorm + ReLU backward with the mask rebuilt from the convolution output (ops.bn_bwd_xmask)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, sink: GradSink, stats):
        B, H, W, C = x.shape
        mom = bn.momentum if bn.momentum is not None else 0.1
        gd = gamma.detach() if gamma is not None else None
        bd = beta.detach() if beta is not None else None
        sync, group = _sync_group(bn)
        total = None
        if sync:   # SyncBatchNorm: local statistics from the partials -> exchange -> coefficients of the GLOBAL ones
            mean_l, var_l = ops.bn_stats(x.view(-1, C), partials=stats)
            mean, rstd, total = sync_batch_stats(mean_l, var_l, B * H * W, bn, group)
            coef = ops.bn_coef_stats(mean, rstd, gd, bd)
        else:
            coef, mean, rstd = ops.bn_coef_partials(stats, B * H * W, gd, bd, bn.running_mean, bn.running_var, mom, bn.eps)
        if bn.num_batches_tracked is not None:
            _count_batch(bn)
        y, arg = ops.maxpool3x3s2_fwd(x.contiguous(), coef, True)
        ctx.save_for_backward(x, gamma, beta, mean, rstd, arg)
        ctx.sink = sink
        ctx.sync = (sync, group, total)
        ctx.wepoch = weights_epoch()   # the backward rebuilds the ReLU mask from gamma / beta (see BatchNormFn)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, rstd, arg = ctx.saved_tensors
        if _bn_params_moved(ctx.wepoch, gamma, beta):
            raise RuntimeError("BatchNorm parameters were updated between this forward and its backward: the ReLU "
                               "mask rebuilt from them would disagree with the forward (set SSL4GIE_STEM_POOL_FUSED=0)")
        B, H, W, C = x.shape
        dz = ops.maxpool3x3s2_bwd(dy.contiguous(), arg, H, W)
        (tg, tb), acc, rets = ctx.sink.plan([gamma, beta])
        gd = gamma.detach() if gamma is not None else None
        bd = beta.detach() if beta is not None else None
        sync, group, total = ctx.sync
        if not sync:
            dx = ops.bn_bwd_xmask(dz.view(-1, C), x.contiguous().view(-1, C), gd, bd, mean, rstd, tg, tb, acc)
            return dx.view(B, H, W, C), rets[0], rets[1], None, None, None
        dz2, x2 = dz.view(-1, C), x.contiguous().view(-1, C)
        sums = ops.bn_bwd_reduce_xmask(dz2, x2, gd, bd, mean, rstd)
        for tgt, row in ((tb, 0), (tg, 1)):  # parameter gradients are the LOCAL sums (DDP averages them)
            if tgt is not None:
                if acc:
                    tgt.add_(sums[row])
                else:
                    tgt.copy_(sums[row])
        gsums = sync_sum(sums, group)
        gsums /= total
        dx = ops.bn_bwd_apply_xmask(dz2, x2, gd, bd, mean, rstd, gsums, 1.0)
        return dx.view(B, H, W, C), rets[0], rets[1], None, None, None


def bn_relu_maxpool_ok(x, bn, stats):
    """whether BnReluMaxPoolFn applies: training-mode statistics from the convolution's partials (under
    SyncBatchNorm: local partials -> exchange -> coefficients of the global statistics), vector-width channels"""
    return (_STEM_POOL_FUSED and stats is not None and (bn.training or bn.running_mean is None)
            and (_SYNC_FUSED or not _sync_group(bn)[0]) and x.shape[-1] % (8 if x.dtype == torch.bfloat16 else 4) == 0)


# SSL4GIE_STEM_POOL_FUSED=0: bn1 / relu / maxpool of the ResNet stem as separate passes again (A/B)
_STEM_POOL_FUSED = __import__("os").environ.get("SSL4GIE_STEM_POOL_FUSED", "1") != "0"


class BnReluConv3x3Fn(torch.autograd.Function):
    """torchvision Bottleneck's bn1 -> relu -> conv2 (3x3, stride 1, no bias) with the BatchNorm applied inside the
    direct convolution kernels: the statistics come from conv1's epilogue partials, the forward and the weight
    gradient normalise (+ ReLU) their halo between its global load and its LDS write, the normalised map is never
    written; the data gradient of conv2 goes through BatchNorm + ReLU backward with the mask rebuilt from the
    BatchNorm input (ops.bn_bwd_xmask).  Values equal BatchNormFn + Conv3x3Fn bit for bit.
    Returns (y, statistics of y or None)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, bn, in_stats, weight, sink: GradSink, lp: LPCache, want_stats):
        from .dpt_engine import _derived
        B, H, W, Cin = x.shape
        dt = x.dtype
        mom = bn.momentum if bn.momentum is not None else 0.1
        coef, mean, rstd = ops.bn_coef_partials(in_stats, B * H * W, gamma.detach() if gamma is not None else None,
                                                beta.detach() if beta is not None else None, bn.running_mean,
                                                bn.running_var, mom, bn.eps)
        if bn.num_batches_tracked is not None:
            _count_batch(bn)
        w2 = _derived(lp, weight, f"c3:{9 * Cin}", dt, lambda w: ops.conv3x3_weight_pack(w, dt, 0, 9 * Cin),
                      recipe=(0, 9 * Cin))
        x = x.contiguous()
        r = ops.conv3x3_direct_fwd(x, w2, None, True, colstats=want_stats, in_coef=coef)
        y, stats = r if want_stats else (r, None)
        ctx.save_for_backward(x, gamma, beta, mean, rstd, coef, weight)
        ctx.cfg = (sink, lp)
        ctx.wepoch = weights_epoch()   # dx rebuilds the ReLU mask from gamma / beta, dW uses the saved `coef`
        if stats is not None:
            ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)
        return y, stats

    @staticmethod
    def backward(ctx, dy, *unused):
        if dy is None:
            return (None,) * 9
        from .dpt_engine import _derived, _write_grad
        x, gamma, beta, mean, rstd, coef, weight = ctx.saved_tensors
        sink, lp = ctx.cfg
        if _bn_params_moved(ctx.wepoch, gamma, beta):
            raise RuntimeError("BatchNorm parameters were updated between this forward and its backward: the ReLU "
                               "mask rebuilt from them would disagree with the forward (set SSL4GIE_BN_CONV_FUSED=0)")
        B, H, W, Cin = x.shape
        Cout = weight.shape[0]
        dt = x.dtype
        dy = dy.contiguous()
        (tw, tg, tb), acc, rets = sink.plan([weight, gamma, beta])
        if tw is not None:  # dW against the normalised operand, rebuilt on the way in
            dw2 = ops.conv3x3_direct_wgrad(dy.view(-1, Cout), x, True, in_coef=coef)
            if tw.is_contiguous():
                ops.conv3x3_wgrad_unpack(dw2, tw, acc)
            else:
                _write_grad(tw, dw2.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2), acc)
        dx = None
        if ctx.needs_input_grad[0]:
            wdd = _derived(lp, weight, "c3dd", dt, lambda w: ops.conv3x3_weight_pack(w, dt, 1), recipe=(1, None))
            dz = ops.conv3x3_direct_fwd(dy.view(B, H, W, Cout), wdd, None)   # gradient at the ReLU's output
            dx = ops.bn_bwd_xmask(dz.view(-1, Cin), x.view(-1, Cin), gamma.detach() if gamma is not None else None,
                                  beta.detach() if beta is not None else None, mean, rstd, tg, tb, acc).view(B, H, W, Cin)
        return dx, rets[1], rets[2], None, None, rets[0], None, None, None


def bn_relu_conv3x3_ok(x, bn, in_stats, conv, need_grad):
    """whether BnReluConv3x3Fn applies: bf16 map on the direct kernels (forward, data gradient and — when a
    backward pass will run — weight gradient), stride 1, no bias, training-mode statistics from the producing
    GEMM's partials, one process (SyncBatchNorm exchanges between statistics and apply)"""
    from .dpt_engine import _DIRECT
    if not (_BN_CONV_FUSED and _DIRECT and in_stats is not None and conv.bias is None and conv.stride[0] == 1):
        return False
    if not (bn.training or bn.running_mean is None) or _sync_group(bn)[0]:
        return False
    Cout, Cin = conv.weight.shape[:2]
    # Cout <= 128 (layer1 / layer2): a workgroup of the direct kernel owns 64 couts and stages the whole halo, so
    # Cout / 64 workgroups would each normalise the same halo again — measured at 256 / 512 couts the kernels lose
    # more (+20 us per launch) than the BatchNorm pass costs (profiles/r04at)
    if x.dtype != torch.bfloat16 or Cin % 64 != 0 or Cout > 128 or not ops.conv3x3_direct_ok(x.contiguous(), Cout):
        return False
    if need_grad:
        B, H, W, _ = x.shape
        dy_like = torch.empty((B, H, W, Cout), dtype=x.dtype, device="meta")  # shape / dtype only
        return ops.conv3x3_direct_wgrad_ok(x.contiguous(), Cout) and ops.conv3x3_direct_ok(dy_like, Cin)
    return True


# SSL4GIE_BN_CONV_FUSED=1 (opt-in): bn1 / relu applied inside the Bottleneck's 3x3 convolution.  Correct (bit-identical,
# tests/test_gpu_resnet.py) and measured NULL on the MoCo-R50 step (62.02 vs 61.96 ms, profiles/r04av): the
# BatchNorm pass it removes is pure HBM streaming (19 us per launch on these narrow maps), while the normalisation
# inside the halo staging is VALU work serialised with the MFMA phases of a kernel that is LDS- / issue-bound
# (+11 .. +21 us per launch, forward and weight gradient) — so the default stays the separate pass.
_BN_CONV_FUSED = __import__("os").environ.get("SSL4GIE_BN_CONV_FUSED", "0") == "1"


class AvgPoolFn(torch.autograd.Function):
    """AdaptiveAvgPool2d(1) + flatten -> fp32 [B, C]"""

    @staticmethod
    def forward(ctx, x):
        ctx.cfg = (x.shape[1], x.shape[2], x.dtype)
        return ops.avgpool_fwd(x.contiguous())

    @staticmethod
    def backward(ctx, dy):
        H, W, dt = ctx.cfg
        return ops.avgpool_bwd(dy.contiguous().float(), H, W, dt)
