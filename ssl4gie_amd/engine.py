"""Host-side execution engine: parameter arena, operand-copy cache, and the autograd glue that
hands whole sub-graphs (a stack of transformer blocks, patch embedding, token assembly, loss) to
the native HIP library.  torch supplies device memory, streams and the autograd tape only.

Precision modes (module attribute `engine_dtype`):
  torch.bfloat16  production: bf16 MFMA operands, fp32 accumulate, fp32 residual stream / LN /
                  softmax statistics  (the reference runs fp16 autocast: SURVEY Appendix E)
  torch.float32   parity: every product on the exact-fp32 MFMA path; matches the reference's CPU
                  fp32 results to ~1e-5 and is what the golden-vector tests run.
"""
from __future__ import annotations

import ctypes as C
import contextlib
import os
import weakref
from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import BlockAct, BlockDims, BlockGrads, BlockWeights


def default_dtype():
    v = os.environ.get("SSL4GIE_PRECISION", "bf16").lower()
    return torch.float32 if v in ("fp32", "f32", "float32") else torch.bfloat16


def _align(n, a=256):
    return (n + a - 1) // a * a


# ------------------------------------------------------------------------------------------------
# parameter / gradient arena
# ------------------------------------------------------------------------------------------------
class ParamArena:
    """All parameters of a model as views of ONE flat fp32 buffer (registration order) and all
    gradients as views of a second one: gradient buckets for the data-parallel all-reduce are
    contiguous slices, no flatten/unflatten copies (see parallel.py)."""

    def __init__(self, params: Sequence[nn.Parameter]):
        params = [p for p in params]
        assert params, "no parameters"
        dev = params[0].device
        self.params = params
        self.offsets = []
        off = 0
        for p in params:
            assert p.dtype == torch.float32 and p.device == dev
            self.offsets.append(off)
            off += _align(p.numel(), 64)  # 256-byte aligned slices
        self.numel = off
        self.data = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self._index = {}
        with torch.no_grad():
            for p, o in zip(params, self.offsets):
                v = self._shaped(self.data[o:o + p.numel()], p)
                v.copy_(p.data)
                p.data = v
                self._index[id(p)] = o
        self.device = dev

    @staticmethod
    def _shaped(flat, p):
        """the slice as a tensor of p's shape; a (C, H, W) parameter stored channels-last (the map LayerNorm
        tables of ViTDet_FPN) keeps that memory order: strides (1, W C, C).  Recognised by its strides, which
        survive .to() / deepcopy / load_state_dict."""
        if p.ndim == 3 and not p.is_contiguous() and p.permute(1, 2, 0).is_contiguous():
            C, H, W = p.shape
            return flat.view(H, W, C).permute(2, 0, 1)
        return flat.view(p.shape)

    def owns(self, p) -> bool:
        o = self._index.get(id(p))
        return o is not None and p.data_ptr() == self.data.data_ptr() + 4 * o

    def intact(self) -> bool:
        return all(self.owns(p) for p in self.params)

    def grad_view(self, p) -> torch.Tensor:
        o = self._index[id(p)]
        return self._shaped(self.grad[o:o + p.numel()], p)

    # -- operand-type (bf16) shadow of the whole arena, refreshed in two launches ---------------
    def lp_views(self, p, dtype):
        """(W [out, in], W^T [in, out]) views of the bf16 shadow arenas for parameter `p` (ndim >= 2),
        refreshing BOTH shadows for every parameter when the weights changed since the last refresh
        (engine.weights_epoch): one flat cast + one batched transposed cast instead of two small
        launches per weight."""
        assert dtype == torch.bfloat16
        if getattr(self, "_lp_stamp", None) != weights_epoch():
            self._refresh_lp()
        o = self._index[id(p)]
        rows = p.shape[0]
        cols = p.numel() // rows
        return (self._lp[o:o + p.numel()].view(rows, cols), self._lpt[o:o + p.numel()].view(cols, rows))

    def lp_flat_for_update(self):
        """the flat bf16 shadow, for an optimizer kernel that writes the updated elements into it
        (optim.ArenaAdamW); brought up to date first, so that segments the kernel skips stay valid"""
        if getattr(self, "_lp_stamp", None) != weights_epoch():
            self._refresh_lp()
        return self._lp

    def lp_flat_is_current(self):
        """called after bump_weights_epoch() by an optimizer that wrote the flat shadow itself.  The
        claim only holds while nobody edits a parameter through torch afterwards (load_state_dict,
        p.copy_, re-initialisation move Tensor._version, not the epoch): the versions are recorded and
        _refresh_lp() trusts the flat shadow only if none has moved since."""
        self._lp_flat_stamp = weights_epoch()
        self._lp_flat_versions = [p._version for p in self.params]

    def _refresh_lp(self):
        L = _lib.load()
        if getattr(self, "_lp", None) is None:
            self._lp = torch.empty(self.numel, dtype=torch.bfloat16, device=self.device)
            self._lpt = torch.zeros(self.numel, dtype=torch.bfloat16, device=self.device)
            mats = [(o, p.shape[0], p.numel() // p.shape[0]) for p, o in zip(self.params, self.offsets)
                    if p.ndim >= 2]
            starts = [0]
            for _, r, c in mats:
                starts.append(starts[-1] + ((r + 63) // 64) * ((c + 63) // 64))
            dev = self.device
            self._lp_tab = (torch.tensor([m[0] for m in mats], dtype=torch.int64, device=dev),
                            torch.tensor([m[1] for m in mats], dtype=torch.int32, device=dev),
                            torch.tensor([m[2] for m in mats], dtype=torch.int32, device=dev),
                            torch.tensor(starts, dtype=torch.int32, device=dev), len(mats), starts[-1])
        flat_ok = getattr(self, "_lp_flat_stamp", None) == weights_epoch() and \
            getattr(self, "_lp_flat_versions", None) == [p._version for p in self.params]
        if not flat_ok:
            _lib.check(L.ssl4gie_cast(ops.ptr(self.data), ops.ptr(self._lp), _lib.BF16, self.numel,
                                      ops.stream()), "cast(arena)")
        off, rows, cols, ts, S, total = self._lp_tab
        if S:
            _lib.check(L.ssl4gie_cast_transpose_batch(ops.ptr(self.data), ops.ptr(self._lpt), ops.ptr(off),
                                                      ops.ptr(rows), ops.ptr(cols), ops.ptr(ts), S, total,
                                                      ops.stream()), "cast_transpose_batch")
        self._lp_stamp = weights_epoch()
        self._lp_versions = {id(p): p._version for p in self.params}

    def span(self, params: Sequence[nn.Parameter]):
        """[start, end) element range of the grad arena covering `params`."""
        os_ = [self._index[id(p)] for p in params]
        ends = [self._index[id(p)] + _align(p.numel(), 64) for p in params]
        return min(os_), max(ends)


# ------------------------------------------------------------------------------------------------
# operand-type weight copies
# ------------------------------------------------------------------------------------------------
# Parameters updated by raw-pointer kernels (ssl4gie_amd.optim) do not bump torch's version counters:
# those optimizers advance this epoch instead, and every operand cache keys on it.
_WEIGHTS_EPOCH = [0]


def weights_epoch() -> int:
    return _WEIGHTS_EPOCH[0]


# epoch -> ids of the parameters that moved when it was entered, or None for "any of them" (an optimizer step);
# lets a cache that holds derived images of MANY parameters re-stamp the untouched ones instead of rebuilding
# them (dpt_engine.refresh_conv_operands: MoCo's EMA update moves the momentum encoder only).  Short history.
_EPOCH_TOUCHED = {}


def bump_weights_epoch(touched=None):
    """`touched`: the parameters whose storage was just rewritten (None = unknown / all)"""
    _WEIGHTS_EPOCH[0] += 1
    _EPOCH_TOUCHED[_WEIGHTS_EPOCH[0]] = None if touched is None else frozenset(id(p) for p in touched)
    for e in [e for e in _EPOCH_TOUCHED if e < _WEIGHTS_EPOCH[0] - 16]:
        del _EPOCH_TOUCHED[e]


def touched_since(epoch: int):
    """ids of the parameters rewritten after `epoch`, or None when that is not known"""
    out = set()
    for e in range(epoch + 1, _WEIGHTS_EPOCH[0] + 1):
        t = _EPOCH_TOUCHED.get(e, None)
        if t is None:
            return None
        out |= t
    return out


# torch's FUSED optimizers (torch.optim.AdamW(..., fused=True)) update parameters without bumping
# Tensor._version, so version counters alone would leave every bf16 operand copy stale after the first
# step (the linear layers would keep computing with their initial weights).  Every optimizer step —
# any optimizer, fused or not — therefore advances the epoch.
try:
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_post_hook
    # `touched`: the parameters this optimizer owns — a step of ANOTHER optimizer (a probe, a second model) between a
    # BatchNorm forward and its backward must not trip the stale-parameter guards (resnet_engine._bn_params_moved)
    _reg_post_hook(lambda opt, _args, _kwargs: bump_weights_epoch(
        touched=[p for g in opt.param_groups for p in g["params"]]))
except ImportError:  # torch < 2.0: callers must use ssl4gie_amd.optim or call bump_weights_epoch()
    pass


class LPCache:
    """bf16 copies W[out,in] and W^T[in,out] of fp32 master weights, refreshed when the parameter's
    version counter or storage changes (optimizer.step / load_state_dict bump it)."""

    def __init__(self, arena_fn=None):
        self._c = {}
        self._arena_fn = arena_fn  # -> ParamArena whose shadow copies serve its own parameters

    def get(self, p: torch.Tensor, dtype, need_t: bool = True):
        w2d = p.detach()
        w2d = w2d.view(w2d.shape[0], -1)
        if dtype == torch.float32:
            return w2d, None
        key = id(p)
        ver = (p._version, p.data_ptr(), dtype, weights_epoch())
        ent = self._c.get(key)
        if ent is None or ent[0] != ver or ent[3]() is not p:
            a = self._arena_fn() if self._arena_fn is not None else None
            if a is not None and dtype == torch.bfloat16 and a.owns(p):
                # the whole arena is refreshed at once (two launches per optimizer step); a parameter
                # edited in place since then (its version moved, the epoch did not) is re-cast alone
                w, wt = a.lp_views(p, dtype)
                if a._lp_versions.get(key) != p._version:
                    ops.cast(w2d, dtype, out=w)
                    ops.cast_transpose(w2d, dtype, out=wt)
                    a._lp_versions[key] = p._version
            else:
                w = ops.cast(w2d, dtype)
                wt = ops.cast_transpose(w2d, dtype)
            ent = (ver, w, wt, weakref.ref(p))
            self._c[key] = ent
        return ent[1], ent[2]


_GLOBAL_LP = LPCache()


# ------------------------------------------------------------------------------------------------
# gradient sinks: where a backward writes parameter gradients
# ------------------------------------------------------------------------------------------------
class GradSink:
    """Decides, per backward call, where parameter gradients are written and what is handed back
    to autograd.  With an arena: kernels write straight into the arena slice and the slice view is
    returned (autograd then just points p.grad at it); if p.grad already aliases the slice
    (gradient accumulation) kernels accumulate in place and None is returned."""

    def __init__(self, arena: Optional[ParamArena]):
        self.arena = arena
        # parameters whose arena slice has been written in the current backward pass while p.grad is
        # still None — i.e. autograd is holding the slice view as their (so far only) contribution
        self._written = set()
        # optional observer (parallel.DataParallel): called with the parameters of ONE use whose
        # gradient kernels have just been enqueued — lets buckets leave per transformer block
        # instead of per autograd node (a whole block stack is a single node)
        self.tracker = None

    def note_done(self, params):
        if self.tracker is not None:
            self.tracker(params)

    def new_pass(self):
        """called at the top of every forward: whatever follows belongs to a new backward pass"""
        self._written.clear()

    def plan(self, params: Sequence[Optional[torch.Tensor]]):
        """-> (targets, accumulate, returns) ; targets[i] is the tensor the kernel writes.

        A parameter used more than once in one forward (both views of MoCo / Barlow Twins go
        through the same encoder and heads) is seen here once per use within a single backward
        pass, with p.grad still None: the first use writes the arena slice and hands the slice view
        to autograd, every later use ACCUMULATES into the slice in place and contributes None —
        autograd holds the first contribution by reference, so the leaf ends up with the sum."""
        live = [p for p in params if p is not None and p.requires_grad]
        arena = self.arena
        if arena is not None and all(arena.owns(p) for p in live):
            views = {id(p): arena.grad_view(p) for p in live}
            tg = [views[id(p)] if (p is not None and p.requires_grad) else None for p in params]
            has = [p.grad is not None for p in live]
            if live and all(has):
                if not all(p.grad.data_ptr() == views[id(p)].data_ptr() for p in live):
                    raise RuntimeError("p.grad was replaced by a tensor outside the gradient arena")
                return tg, True, [None] * len(params)
            if any(has):
                raise RuntimeError("mixed .grad state (some None, some set) is not supported")
            seen = [id(p) in self._written for p in live]
            if live and all(seen):
                return tg, True, [None] * len(params)
            if any(seen):
                raise RuntimeError("parameters of one node were written by different earlier nodes")
            self._written.update(id(p) for p in live)
            return tg, False, list(tg)
        tg = [torch.empty_like(p) if (p is not None and p.requires_grad) else None for p in params]
        return tg, False, list(tg)


# ------------------------------------------------------------------------------------------------
# transformer-block stack
# ------------------------------------------------------------------------------------------------
BLOCK_PARAM_ORDER = ("norm1.weight", "norm1.bias", "attn.qkv.weight", "attn.qkv.bias",
                     "attn.proj.weight", "attn.proj.bias", "norm2.weight", "norm2.bias",
                     "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")
_NP = len(BLOCK_PARAM_ORDER)


@dataclass
class StackCfg:
    heads: int
    eps: float
    dtype: torch.dtype
    taps: tuple            # block indices whose outputs are returned as well
    need_bwd: bool
    sink: GradSink
    lp: LPCache
    on_block_grads: Optional[Callable[[int], None]] = None  # optional per-block call-back (unused by
    # parallel.DataParallel, which takes readiness from autograd's post-accumulate-grad hooks)


def _block_params(blk):
    return [blk.norm1.weight, blk.norm1.bias, blk.attn.qkv.weight, blk.attn.qkv.bias,
            blk.attn.proj.weight, blk.attn.proj.bias, blk.norm2.weight, blk.norm2.bias,
            blk.mlp.fc1.weight, blk.mlp.fc1.bias, blk.mlp.fc2.weight, blk.mlp.fc2.bias]


def _weights_struct(ps, cfg: StackCfg) -> tuple:
    n1w, n1b, wqkv, bqkv, wproj, bproj, n2w, n2b, wfc1, bfc1, wfc2, bfc2 = ps
    keep = []
    w = BlockWeights()
    w.ln1_g, w.ln1_b, w.ln2_g, w.ln2_b = (ops.ptr(t) for t in (n1w, n1b, n2w, n2b))
    w.bqkv, w.bproj, w.bfc1, w.bfc2 = (ops.ptr(t) for t in (bqkv, bproj, bfc1, bfc2))
    for name, p in (("wqkv", wqkv), ("wproj", wproj), ("wfc1", wfc1), ("wfc2", wfc2)):
        a, at = cfg.lp.get(p, cfg.dtype)
        keep += [a, at]
        setattr(w, name, ops.ptr(a))
        setattr(w, name + "_t", ops.ptr(at))
    return w, keep


class _ActArena:
    """Saved activations of one block, carved out of one byte buffer."""

    def __init__(self, B, N, D, H, F, es):
        T = B * N
        self.sizes = [("mean1", T * 4), ("rstd1", T * 4), ("mean2", T * 4), ("rstd2", T * 4),
                      ("h1", T * D * es), ("qkv", T * 3 * D * es), ("attn", T * D * es),
                      ("lse", B * H * N * 4), ("xmid", T * D * 4), ("h2", T * D * es),
                      ("u", T * F * es), ("g", T * F * es)]
        self.total = sum(_align(s) for _, s in self.sizes)

    def struct(self, base_ptr: int) -> BlockAct:
        a = BlockAct()
        off = 0
        for name, s in self.sizes:
            setattr(a, name, base_ptr + off)
            off += _align(s)
        return a


class BlockStackFn(torch.autograd.Function):
    """x -> blocks[depth-1](...blocks[0](x)); also returns the tap outputs.  inputs: x, cfg,
    12*depth parameters in BLOCK_PARAM_ORDER."""

    @staticmethod
    def forward(ctx, x, cfg: StackCfg, *params):
        L = _lib.load()
        depth = len(params) // _NP
        B, N, D = x.shape
        F = params[8].shape[0]
        assert x.dtype == torch.float32 and x.is_cuda
        assert D % cfg.heads == 0 and params[2].shape == (3 * D, D)
        x = x.contiguous()
        es = 2 if cfg.dtype == torch.bfloat16 else 4
        dims = BlockDims(B, N, D, cfg.heads, F, ops.code(cfg.dtype), float(cfg.eps))
        wsb = L.ssl4gie_block_workspace_bytes(C.byref(dims))
        assert wsb > 0
        ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
        layout = _ActArena(B, N, D, cfg.heads, F, es)
        nact = depth if cfg.need_bwd else 1
        acts = torch.empty(nact * layout.total, dtype=torch.uint8, device=x.device)
        xs = [x]
        keepalive = []
        wstructs = []
        st = ops.stream()
        for i in range(depth):
            w, keep = _weights_struct(params[i * _NP:(i + 1) * _NP], cfg)
            keepalive += keep
            wstructs.append(w)
            a = layout.struct(acts.data_ptr() + (i if cfg.need_bwd else 0) * layout.total)
            xo = torch.empty_like(x)
            _lib.check(L.ssl4gie_block_fwd(C.byref(dims), C.byref(w), C.byref(a), xs[-1].data_ptr(),
                                           xo.data_ptr(), ws.data_ptr(), st), f"block_fwd[{i}]")
            xs.append(xo)
        outs = [xs[depth]] + [xs[t + 1] for t in cfg.taps]
        if cfg.need_bwd:
            ctx.cfg, ctx.dims, ctx.layout = cfg, dims, layout
            ctx.acts, ctx.xs, ctx.ws = acts, xs, ws
            ctx.wstructs, ctx.keepalive = wstructs, keepalive
            ctx.params = params
            ctx.depth = depth
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        L = _lib.load()
        cfg, dims, layout, depth = ctx.cfg, ctx.dims, ctx.layout, ctx.depth
        x0 = ctx.xs[0]
        lp = cfg.dtype if cfg.dtype != torch.float32 else None
        tap_g = {t: g for t, g in zip(cfg.taps, gouts[1:]) if g is not None}
        st = ops.stream()
        targets, accumulate, returns = cfg.sink.plan(ctx.params)
        dx = gouts[0]
        if dx is None:
            dx = torch.zeros_like(x0)
        dx = dx.contiguous()
        pending_tap = tap_g.pop(depth - 1, None)
        dx, dx_lp = ops.add_cast(dx, pending_tap, want_f32=True, lp_dtype=lp)
        # Deferred weight gradients (bf16 engine, opt-in: _wgrad_group_size): a block's four dW
        # products only have 48-108 output tiles, so alone (or in pairs) they must split K over the
        # CUs and pay for fp32 slabs and their reduction.  Here the products of `grp` consecutive
        # blocks are collected and run as ONE grouped launch with ~200 whole-K tiles — no slabs at
        # all — on the weight-gradient side stream, beside the next blocks' data-gradient chain.
        # Every block of a group keeps its own workspace (the products' dY operands live there)
        # from a ring of 2 groups.
        grp = _wgrad_group_size(dims, depth) if lp is not None else 0
        if grp:
            ring = [ctx.ws] + [torch.empty_like(ctx.ws) for _ in range(2 * grp - 1)]
            descs = (_lib.GemmDesc * (4 * grp))()
            gws_bytes, gws = 0, [None, None]
            pending, keep, n_groups = [], [[], []], 0
            flags = int(accumulate) | _lib.BWD_DEFER_WGRAD

            def launch_group():
                nonlocal n_groups, gws_bytes
                n = 4 * len(pending)
                slot = n_groups % 2
                need = L.ssl4gie_gemm_tn_group_workspace_bytes(descs, n)
                if need and (gws[slot] is None or gws[slot].numel() < need):
                    gws[slot] = torch.empty(need, dtype=torch.uint8, device=x0.device)
                _lib.check(L.ssl4gie_wgrad_group(descs, n, ops.ptr(gws[slot]), need, slot, st), "wgrad_group")
                n_groups += 1
                done = list(pending)
                pending.clear()
                if cfg.on_block_grads is not None:
                    for j in done:
                        cfg.on_block_grads(j)
        nojoin = (not grp) and lp is not None and _BWD_NO_JOIN and depth > 1
        if nojoin:
            ring2 = [ctx.ws, torch.empty_like(ctx.ws)]
            keep2 = [[], []]
        for i in range(depth - 1, -1, -1):
            if i != depth - 1 and i in tap_g:
                dx, dx_lp = ops.add_cast(dx, tap_g[i].contiguous(), want_f32=True, lp_dtype=lp)
            g = BlockGrads()
            tg = targets[i * _NP:(i + 1) * _NP]
            scratch = []
            for name, t, p in zip(("ln1_g", "ln1_b", "wqkv", "bqkv", "wproj", "bproj", "ln2_g",
                                   "ln2_b", "wfc1", "bfc1", "wfc2", "bfc2"), tg,
                                  ctx.params[i * _NP:(i + 1) * _NP]):
                if t is None:  # frozen parameter: the executor still needs somewhere to write
                    t = torch.empty_like(p)
                    scratch.append(t)
                setattr(g, name, t.data_ptr())
            a = layout.struct(ctx.acts.data_ptr() + i * layout.total)
            dxn = torch.empty_like(x0)
            dxn_lp = torch.empty(x0.shape, dtype=lp, device=x0.device) if lp else None
            if grp:
                k = depth - 1 - i                      # blocks done so far
                if k >= 2 * grp and k % grp == 0:      # about to reuse the workspaces of group k/grp - 2
                    _lib.check(L.ssl4gie_wgrad_wait((k // grp) % 2, st), "wgrad_wait")
                    keep[(k // grp) % 2].clear()
                ws_i = ring[k % (2 * grp)]
                _lib.check(L.ssl4gie_block_bwd(C.byref(dims), C.byref(ctx.wstructs[i]), C.byref(a),
                                               C.byref(g), ctx.xs[i].data_ptr(), dx.data_ptr(),
                                               ops.ptr(dx_lp), dxn.data_ptr(), ops.ptr(dxn_lp),
                                               flags, ws_i.data_ptr(), st), f"block_bwd[{i}]")
                _lib.check(L.ssl4gie_block_wgrad_descs(C.byref(dims), C.byref(a), C.byref(g), ops.ptr(dx_lp),
                                                       ws_i.data_ptr(), int(accumulate),
                                                       C.byref(descs, 4 * len(pending) * C.sizeof(_lib.GemmDesc))),
                           f"block_wgrad_descs[{i}]")
                pending.append(i)
                keep[(k // grp) % 2] += [dx_lp, scratch]  # dY of dW_fc2 (+ frozen-parameter scratch)
                if len(pending) == grp or i == 0:
                    launch_group()
            elif nojoin:
                # two workspaces in alternation: the caller's stream does not wait for this block's weight
                # gradients (side stream) — only, two blocks later, before their operands are overwritten
                k = (depth - 1 - i) % 2
                _lib.check(L.ssl4gie_wgrad_wait(k, st), "wgrad_wait")
                keep2[k].clear()
                _lib.check(L.ssl4gie_block_bwd(C.byref(dims), C.byref(ctx.wstructs[i]), C.byref(a),
                                               C.byref(g), ctx.xs[i].data_ptr(), dx.data_ptr(),
                                               ops.ptr(dx_lp), dxn.data_ptr(), ops.ptr(dxn_lp),
                                               int(accumulate) | _lib.BWD_NO_JOIN | (k << 4), ring2[k].data_ptr(), st),
                           f"block_bwd[{i}]")
                keep2[k] += [dx, dx_lp, scratch]   # dY of dW_fc2 / frozen-parameter scratch: read on the side stream
                if cfg.on_block_grads is not None:
                    cfg.on_block_grads(i)
            else:
                _lib.check(L.ssl4gie_block_bwd(C.byref(dims), C.byref(ctx.wstructs[i]), C.byref(a),
                                               C.byref(g), ctx.xs[i].data_ptr(), dx.data_ptr(),
                                               ops.ptr(dx_lp), dxn.data_ptr(), ops.ptr(dxn_lp),
                                               int(accumulate), ctx.ws.data_ptr(), st),
                           f"block_bwd[{i}]")
                if cfg.on_block_grads is not None:
                    cfg.on_block_grads(i)
            dx, dx_lp = dxn, dxn_lp
        if grp:  # the caller's stream continues after the last weight gradient; operands may go
            for slot in range(min(n_groups, 2)):
                _lib.check(L.ssl4gie_wgrad_wait(slot, st), "wgrad_wait")
            keep[0].clear(); keep[1].clear()
        if nojoin:
            for slot in (0, 1):
                _lib.check(L.ssl4gie_wgrad_wait(slot, st), "wgrad_wait")
            keep2[0].clear(); keep2[1].clear()
        ctx.acts = ctx.xs = ctx.ws = ctx.keepalive = None
        return (dx, None) + tuple(returns)


# SSL4GIE_BWD_NO_JOIN=0: every block's backward ends with the caller's stream waiting for its weight gradients
# again (one workspace) — A/B
_BWD_NO_JOIN = os.environ.get("SSL4GIE_BWD_NO_JOIN", "1") != "0"


def _wgrad_group_size(dims, depth: int) -> int:
    """blocks per grouped weight-gradient launch (SSL4GIE_WGRAD_GROUP = 1..4); 0 / unset = the
    per-block paired launches.  Measured on the MAE ViT-B step (profiles/r02b_wgrad_group_ab.log):
    with everything on one stream the grouped launches save 0.8 ms of slab traffic (25.55 -> 24.75 ms),
    but beside the data-gradient chain the per-block pairs fill the CUs the chain leaves idle far
    better (24.25 ms) than a burst of ~200 long workgroups every few blocks (24.6 ms) — so the
    pairs stay the default and the groups are an option for single-stream runs."""
    env = os.environ.get("SSL4GIE_WGRAD_GROUP", "").strip()
    if not env:
        return 0
    D, F = dims.D, dims.F
    if D % 4 or F % 4:
        return 0
    return max(0, min(int(env), 4, depth))


def run_blocks(blocks: Sequence[nn.Module], x: torch.Tensor, heads: int, eps: float, dtype,
               sink: GradSink, taps: Sequence[int] = (), lp: LPCache = None,
               on_block_grads=None):
    """Run a list of Block parameter-holders; returns (x_out, [tap outputs])."""
    params: List[torch.Tensor] = []
    for b in blocks:
        params += _block_params(b)
    need = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
    cfg = StackCfg(heads=heads, eps=eps, dtype=dtype, taps=tuple(taps), need_bwd=need, sink=sink,
                   lp=lp or _GLOBAL_LP, on_block_grads=on_block_grads)
    outs = BlockStackFn.apply(x, cfg, *params)
    return outs[0], list(outs[1:])


# ------------------------------------------------------------------------------------------------
# single-op autograd nodes
# ------------------------------------------------------------------------------------------------
class LayerNormFn(torch.autograd.Function):
    """fp32 residual stream -> normalised rows in the MFMA operand type."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, out_dtype, sink):
        x = x.contiguous()
        y, mean, rstd = ops.layernorm_fwd(x, gamma.detach(), beta.detach(), eps, out_dtype)
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.sink = sink
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        (tg, tb), acc, rets = ctx.sink.plan([gamma, beta])
        if tg is None or tb is None:  # frozen affine: scratch
            tg = tg if tg is not None else torch.empty_like(gamma)
            tb = tb if tb is not None else torch.empty_like(beta)
        dx, _, _, _ = ops.layernorm_bwd(dy.contiguous(), x, gamma.detach(), mean, rstd,
                                        dgamma=tg, dbeta=tb, accumulate=acc)
        return dx, rets[0], rets[1], None, None, None


class GradJoin:
    """Joins the gradient contributions of a tensor that feeds several engine nodes of one block
    (ResNet bottleneck input: conv1 + identity / downsample branch) without autograd's separate add
    kernel.  Nodes are handed the same GradJoin in forward, in creation order; backward runs them in
    reverse, so the FIRST node created is the last to run: it is the `adder` and takes everything the
    others deposited into its data-gradient GEMM's epilogue (ssl4gie EPI_ADD_AUX); the others return
    None for that input.  A join is only created when the tensor requires grad, so the adder's
    backward is guaranteed to run."""

    __slots__ = ("adder_claimed", "pending")

    def __init__(self):
        self.adder_claimed = False
        self.pending = None

    @staticmethod
    def for_tensor(x):
        return GradJoin() if (torch.is_grad_enabled() and x.requires_grad) else None

    def claim(self) -> str:
        if not self.adder_claimed:
            self.adder_claimed = True
            return "adder"
        return "depositor"

    def deposit(self, g):
        self.pending = g if self.pending is None else self.pending + g

    def take(self):
        g, self.pending = self.pending, None
        return g


# ---- weight gradients of single layers beside their data gradients -----------------------------------------------
# (SSL4GIE_CONV_WGRAD_STREAM=0 turns it off.)  LinearFn / Conv3x3Fn.backward enqueue their weight-gradient product (and what follows
# it: slab reduction, scatter into the parameter's layout) on a second stream forked from the caller's, so that it
# runs beside the data-gradient chain — what the transformer blocks' executor does with the library's side stream.
# The caller's stream joins at the end of the backward pass (autograd-engine call-backs); operands are kept alive
# until then.  DataParallel's communication stream waits for this stream before every bucket (parallel._wait_wgrad_stream).
_WGRAD_SIDE = os.environ.get("SSL4GIE_CONV_WGRAD_STREAM", "1") != "0"
_WG = {"stream": None}


def set_wgrad_side(on: bool) -> bool:
    """turn the single layers' weight-gradient stream on / off at run time (bench.py folds every stream back while it
    takes per-kernel durations); returns the previous setting"""
    global _WGRAD_SIDE
    prev, _WGRAD_SIDE = _WGRAD_SIDE, bool(on)
    return prev


SAFE_POST_ACC_HOOK_IDS = set()   # handle ids of post-accumulate-grad hooks that order themselves after the side stream


def _wgrad_join():
    if _WG["stream"] is not None:
        torch.cuda.current_stream().wait_stream(_WG["stream"])


def wgrad_fork(sink, params, *operands):
    """-> the weight-gradient stream (forked from the current one) or None.

    The fork is taken only when nothing can read the gradient tensors on the caller's stream before the join at the
    end of the backward pass: the targets must be views of the gradient arena (autograd's AccumulateGrad
    then keeps the returned view by reference — no kernel), the parameters must carry no tensor hooks (a hook reads
    the gradient at once, on the caller's stream), and the pass must not build a graph (create_graph differentiates
    through the returned tensors).  Otherwise the product runs on the caller's stream as before.  `operands` are
    handed to the allocator with record_stream: their memory is not reused before the side stream has passed this
    point, and each layer's tensors can be released as soon as its node is done (no list kept until the join)."""
    if not _WGRAD_SIDE or not operands or operands[0] is None or not operands[0].is_cuda:
        return None
    live = [p for p in params if p is not None and p.requires_grad]
    arena = getattr(sink, "arena", None)
    if arena is None or not live or not all(arena.owns(p) for p in live):
        return None
    if torch.is_grad_enabled() or any(getattr(p, "_backward_hooks", None) for p in live):
        return None
    # a post-accumulate-grad hook of the user's (optimizer-in-backward, per-parameter clipping) reads the arena
    # gradient on the caller's stream as soon as the node returns; DataParallel's own hook is safe (its comm stream
    # waits on the side stream) and is recognised by its handle id
    for p in live:
        hooks = getattr(p, "_post_accumulate_grad_hooks", None)
        if hooks and any(k not in SAFE_POST_ACC_HOOK_IDS for k in hooks):
            return None
    if _WG["stream"] is None:
        _WG["stream"] = torch.cuda.Stream()
    side = _WG["stream"]
    side.wait_stream(torch.cuda.current_stream())
    for t in operands:
        if t is not None and t.is_cuda:
            t.record_stream(side)
    from torch.autograd import Variable
    Variable._execution_engine.queue_callback(_wgrad_join)   # idempotent: one per use, all run at the end of the pass
    return side


class LinearFn(torch.autograd.Function):
    """y = x W^T + b on operand-type activations; y in `out_dtype`."""

    @staticmethod
    def forward(ctx, x, weight, bias, dtype, out_dtype, sink, lp, want_stats=False, join=None):
        """want_stats: also return the per-128-row column sums / sums of squares of y (fp32
        [parts, 2, n_out], not differentiable) for the BatchNorm that follows — or None when the
        GEMM cannot produce them (ops.colstats_ok).  join: GradJoin of the input tensor."""
        ctx.join, ctx.join_role = join, (join.claim() if join is not None else None)
        shp = x.shape
        x2 = x.contiguous().view(-1, shp[-1])
        assert x2.dtype == dtype
        w, wt = lp.get(weight, dtype)
        ctx.save_for_backward(x2, weight, bias)
        ctx.wlp, ctx.wt, ctx.sink, ctx.dtype, ctx.shp = w, wt, sink, dtype, shp
        if want_stats:
            stats = None
            if bias is None and out_dtype == dtype and ops.colstats_ok(x2.shape[0], w.shape[0], w.shape[1], dtype):
                y, stats = ops.linear_fwd(x2, w, None, out_dtype=out_dtype, colstats=True)
                ctx.mark_non_differentiable(stats)
                ctx.set_materialize_grads(False)  # no zero tensor for the statistics' (absent) gradient
            else:
                y = ops.linear_fwd(x2, w, bias.detach() if bias is not None else None, out_dtype=out_dtype)
            return y.view(*shp[:-1], w.shape[0]), stats
        y = ops.linear_fwd(x2, w, bias.detach() if bias is not None else None, out_dtype=out_dtype)
        return y.view(*shp[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy, *unused):
        if dy is None:  # the map itself was not used downstream (gradients are not materialised)
            return (None,) * 9
        x2, weight, bias = ctx.saved_tensors
        dy2 = dy.contiguous().view(-1, dy.shape[-1])
        if dy2.dtype != ctx.dtype:  # fp32 output (decoder_pred): operand copy for the GEMMs
            _, dy2 = ops.add_cast(dy2, None, want_f32=False, lp_dtype=ctx.dtype)
        (tw, tb), acc, rets = ctx.sink.plan([weight, bias])
        dx = None
        if ctx.needs_input_grad[0]:
            other = ctx.join.take() if ctx.join_role == "adder" else None
            if other is not None and ops.add_aux_ok(dy2.shape[0], x2.shape[1], dy2.shape[1], ctx.dtype,
                                                    ctx.wt is not None):
                dx = ops.linear_bwd_data(dy2, ctx.wlp, ctx.wt, add_aux=other.contiguous()).view(ctx.shp)
            else:
                dx = ops.linear_bwd_data(dy2, ctx.wlp, ctx.wt).view(ctx.shp)
                if other is not None:
                    dx = dx + other.view(ctx.shp)
            if ctx.join_role == "depositor":
                ctx.join.deposit(dx)
                dx = None
        if tw is not None:
            side = wgrad_fork(ctx.sink, (weight, bias), dy2, x2, weight, bias)   # beside the data-gradient chain when safe
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                ops.linear_bwd_weight(dy2, x2, out=tw, accumulate=acc, bias_out=tb)
        elif tb is not None:
            ops.colsum(dy2, out=tb, accumulate=acc)
        return dx, rets[0], rets[1], None, None, None, None, None, None


class MatmulNTFn(torch.autograd.Function):
    """a @ b^T for two fp32 [rows, C] matrices on the library's GEMM (torch.einsum('nc,mc->nm', [q, k]) of
    MoCo's InfoNCE, moco/builder.py:83; `q @ k.t()` would be two hipBLASLt launches per call and direction)"""

    @staticmethod
    def forward(ctx, a, b):
        a2, b2 = a.contiguous().float(), b.contiguous().float()
        ctx.save_for_backward(a2, b2)
        return ops.linear_fwd(a2, b2, None, out_dtype=torch.float32)

    @staticmethod
    def backward(ctx, dy):
        a2, b2 = ctx.saved_tensors
        dy2 = dy.contiguous().float()
        da = ops.linear_bwd_data(dy2, b2, out_dtype=torch.float32) if ctx.needs_input_grad[0] else None
        db = ops.linear_bwd_weight(dy2, a2) if ctx.needs_input_grad[1] else None
        return da, db


class PatchEmbedFn(torch.autograd.Function):
    """PatchEmbed conv (k=s=p) as gather + GEMM, + pos-embed add + cls concat, computed ONLY for the
    patches that survive masking (`ids[:, :nsel]`): identical per-row arithmetic to embedding all
    patches and gathering afterwards (models_mae.py:152-163), a quarter of the work."""

    @staticmethod
    def forward(ctx, imgs, weight, bias, cls, pos, ids, nsel, p, dtype, sink, lp):
        B = imgs.shape[0]
        imgs = imgs.contiguous()
        cols = ops.patch_gather(imgs, p, ids=ids, nsel=nsel, out_dtype=dtype)
        w, _ = lp.get(weight, dtype)
        y = ops.linear_fwd(cols, w, bias.detach(), out_dtype=dtype)
        nsel_eff = cols.shape[0] // B
        x = ops.tokens_assemble(y, cls.detach().contiguous().view(-1),
                                pos.detach().contiguous().view(-1, pos.shape[-1]), B, nsel_eff,
                                ids=ids)
        ctx.save_for_backward(cols, weight, bias, cls, pos)
        ctx.sink, ctx.dtype = sink, dtype
        # a trainable position table (timm's default; the MAE / MoCo tables are fixed) gets its
        # gradient only on the unmasked path, where row j of every image meets pos[j]
        ctx.pos_grad = pos.requires_grad and ids is None
        if pos.requires_grad and ids is not None:
            raise NotImplementedError("masked patch embedding with a trainable pos_embed")
        return x

    @staticmethod
    def backward(ctx, dx):
        cols, weight, bias, cls, pos = ctx.saved_tensors
        plist = [weight, bias, cls] + ([pos] if ctx.pos_grad else [])
        tgs, acc, rets = ctx.sink.plan(plist)
        tw, tb, tc = tgs[:3]
        dx = dx.contiguous()
        if ctx.pos_grad and tgs[3] is not None:
            dpos = dx.sum(0, keepdim=True)  # [1, 1 + L, D]
            if acc:
                tgs[3].add_(dpos)
            else:
                tgs[3].copy_(dpos)
        dy = ops.tokens_assemble_bwd(dx, ctx.dtype, dcls_out=tc, accumulate=acc)
        if tw is not None:
            ops.linear_bwd_weight(dy, cols, out=tw, accumulate=acc, bias_out=tb)
        elif tb is not None:
            ops.colsum(dy, out=tb, accumulate=acc)
        return (None, rets[0], rets[1], rets[2], rets[3] if ctx.pos_grad else None) + (None,) * 6


class PatchEmbedDetFn(torch.autograd.Function):
    """PatchEmbed conv (k=s=p) of the detection trunk: gather the patches in the order `ids` asks
    for (window-major) + GEMM with bias -> fp32 [B*N, D]; no cls token, the position table is added
    by the caller (reference models.py:326-328 with det=True)."""

    @staticmethod
    def forward(ctx, imgs, weight, bias, ids, p, dtype, sink, lp):
        cols = ops.patch_gather(imgs.contiguous(), p, ids=ids.contiguous(), out_dtype=dtype)
        w, _ = lp.get(weight, dtype)
        y = ops.linear_fwd(cols, w, bias.detach(), out_dtype=torch.float32)
        ctx.save_for_backward(cols, weight, bias)
        ctx.sink, ctx.dtype = sink, dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        cols, weight, bias = ctx.saved_tensors
        (tw, tb), acc, rets = ctx.sink.plan([weight, bias])
        dy = dy.contiguous()
        dyl = dy if ctx.dtype == torch.float32 else ops.cast(dy, ctx.dtype)
        if tw is not None:
            ops.linear_bwd_weight(dyl, cols, out=tw, accumulate=acc, bias_out=tb)
        elif tb is not None:
            ops.colsum(dyl, out=tb, accumulate=acc)
        return (None, rets[0], rets[1]) + (None,) * 5


class DecoderAssembleFn(torch.autograd.Function):
    """append mask tokens, un-shuffle, add decoder pos-embed (models_mae.py:177-183)."""

    @staticmethod
    def forward(ctx, y, mask_token, dpos, ids_restore, ids_shuffle, nkeep, sink):
        xd = ops.decoder_assemble(y.contiguous(), mask_token.detach().contiguous().view(-1),
                                  dpos.detach().contiguous().view(-1, dpos.shape[-1]),
                                  ids_restore, nkeep)
        ctx.save_for_backward(mask_token, ids_shuffle)
        ctx.nkeep, ctx.sink, ctx.ydtype = nkeep, sink, y.dtype
        return xd

    @staticmethod
    def backward(ctx, dxd):
        mask_token, ids_shuffle = ctx.saved_tensors
        (tm,), acc, rets = ctx.sink.plan([mask_token])
        if tm is None:
            tm = torch.empty_like(mask_token)
        dy = ops.decoder_assemble_bwd(dxd.contiguous(), ids_shuffle, ctx.nkeep, ctx.ydtype,
                                      dmask_out=tm, accumulate=acc)
        return dy, rets[0], None, None, None, None, None


class MaeLossFn(torch.autograd.Function):
    """per-patch masked MSE against (optionally normalised) pixels (models_mae.py:198-212)."""

    @staticmethod
    def forward(ctx, pred_full, imgs, mask, p, norm_pix):
        pred_full = pred_full.contiguous()
        per_patch = ops.mae_loss_fwd(pred_full, imgs, mask, p, norm_pix)
        ctx.save_for_backward(pred_full, imgs, mask)
        ctx.p, ctx.norm_pix = p, norm_pix
        return per_patch

    @staticmethod
    def backward(ctx, gpp):
        pred_full, imgs, mask = ctx.saved_tensors
        dpred = ops.mae_loss_bwd(pred_full, imgs, mask, ctx.p, ctx.norm_pix, gpp.contiguous())
        return dpred, None, None, None, None


# ------------------------------------------------------------------------------------------------
# base class for the model mirrors
# ------------------------------------------------------------------------------------------------
class EngineModule(nn.Module):
    """nn.Module whose forward runs on the HIP engine.  Builds the parameter arena lazily on the
    first forward on a device (and again if .to()/.cuda() moved the parameters)."""

    engine_dtype = None  # torch dtype; None -> default_dtype()

    def __init__(self):
        super().__init__()
        self._arena: Optional[ParamArena] = None
        self._lp = LPCache(self._arena_nocheck)
        self._sink = None
        object.__setattr__(self, "_root", None)  # owning EngineModule when nested (not a child link)

    def adopt(self, child: "EngineModule"):
        """A nested EngineModule (e.g. the DPT decoder inside ViT_from_MAE) shares the parent's
        arena, operand cache and precision: parameters live in ONE arena per model."""
        object.__setattr__(child, "_root", self)
        return child

    def set_precision(self, name_or_dtype):
        if isinstance(name_or_dtype, str):
            name_or_dtype = {"bf16": torch.bfloat16, "fp32": torch.float32,
                             "f32": torch.float32}[name_or_dtype.lower()]
        assert name_or_dtype in (torch.bfloat16, torch.float32)
        self.engine_dtype = name_or_dtype
        return self

    @property
    def dtype_(self):
        if self._root is not None:
            return self._root.dtype_
        return self.engine_dtype or default_dtype()

    def arena(self) -> ParamArena:
        if self._root is not None:
            return self._root.arena()
        ps = list(self.parameters())
        a = self._arena
        if a is None or len(a.params) != len(ps) or any(x is not y for x, y in zip(a.params, ps)) \
                or not a.intact():
            if not ps[0].is_cuda:
                raise RuntimeError("ssl4gie_amd models run on the HIP device only: call .cuda() "
                                   "first (there is no CPU fallback)")
            old = {id(p): p.grad for p in ps}
            self._arena = a = ParamArena(ps)
            self._sink = GradSink(a)
            for p in ps:  # gradients that existed before a rebuild are carried over
                g = old[id(p)]
                if g is not None:
                    v = a.grad_view(p)
                    v.copy_(g)
                    p.grad = v
        return a

    def _arena_nocheck(self) -> Optional[ParamArena]:
        """the arena as validated by the last _prepare() (no parameter walk): operand-cache lookups"""
        return self._root._arena_nocheck() if self._root is not None else self._arena

    def _prepare(self):
        """Call once at the top of forward(): validates the arena (cheap pointer checks)."""
        _lib.load()
        self.arena()
        self.sink().new_pass()
        if self.lp_cache.__dict__.get("_conv_recipes"):   # 3x3 operand images: one launch for all stale ones
            from .dpt_engine import refresh_conv_operands
            refresh_conv_operands(self.lp_cache)

    def sink(self) -> GradSink:
        if self._root is not None:
            return self._root.sink()
        if self._arena is None:
            self.arena()
        return self._sink

    @property
    def lp_cache(self) -> LPCache:
        return self._root.lp_cache if self._root is not None else self._lp

    # helpers used by the model mirrors -----------------------------------------------------
    def _ln(self, x, norm: nn.LayerNorm, out_dtype=None):
        return LayerNormFn.apply(x, norm.weight, norm.bias, norm.eps, out_dtype or self.dtype_,
                                 self.sink())

    def _linear(self, x, lin: nn.Linear, out_dtype=None):
        return LinearFn.apply(x, lin.weight, lin.bias, self.dtype_, out_dtype or self.dtype_,
                              self.sink(), self._lp)

    def _blocks(self, blocks, x, heads, eps, taps=()):
        sink = self.sink()
        hook = None
        if sink.tracker is not None:
            blist = list(blocks)
            hook = lambda i: sink.note_done(_block_params(blist[i]))
        return run_blocks(blocks, x, heads, eps, self.dtype_, sink, taps=taps, lp=self._lp,
                          on_block_grads=hook)
