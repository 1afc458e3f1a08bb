"""Optimizer steps as kernels over the parameter arena.

`ArenaAdamW` is torch.optim.AdamW's update (the reference's MAE / depth drivers:
`Models/mae/main_pretrain.py:179-180` with `add_weight_decay` groups, `train_depth.py:280`) and
`ArenaLARS` the reference's `Models/moco_v3/moco/optimizer.py:18-43`, each as one pass (LARS: norm
pass + apply pass) over the flat fp32 buffers of `engine.ParamArena` — instead of a multi-tensor
launch chain per parameter group.  Both take torch-style parameter groups, keep `param_groups`
(so `lr_sched.adjust_learning_rate` and friends work unchanged) and skip parameters that are frozen
or have no gradient in this step, as torch does.

The north_star leaves the optimizer step host-side (torch); these are the optional fused variants.
The engine's operand caches are keyed on torch's version counters, which raw-pointer kernels do not
bump: `step()` advances `engine.weights_epoch` instead.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .engine import bump_weights_epoch
from .ops import ptr, stream


class _ArenaOptimizer:
    def __init__(self, model, params, defaults):
        self.model = model
        if isinstance(params, (list, tuple)) and params and isinstance(params[0], dict):
            groups = [dict(g) for g in params]
        else:
            groups = [{"params": list(params)}]
        for g in groups:
            g["params"] = list(g["params"])
            for k, v in defaults.items():
                g.setdefault(k, v)
        self.param_groups = groups
        self.defaults = dict(defaults)
        self._tables_key = None
        self._hyper = None
        self.step_count = 0

    # ------------------------------------------------------------------ segment tables
    def _arena(self):
        # torch optimizers get this through the global pre-step hook parallel.py registers: close any
        # data-parallel pass still open and wait for the comm stream before gradients are read
        from .parallel import before_optimizer_step
        before_optimizer_step()
        a = self.model.arena()
        own = [p for g in self.param_groups for p in g["params"]]
        assert all(a.owns(p) for p in own), "every optimised parameter must live in the model's arena"
        # a gradient autograd produced with torch ops lives outside the gradient arena the kernels
        # read: move it into the parameter's slice (parallel.DataParallel does the same)
        for p in own:
            if p.grad is not None:
                v = a.grad_view(p)
                if p.grad.data_ptr() != v.data_ptr():
                    v.copy_(p.grad)
                    p.grad = v
        return a

    def _build_tables(self, a):
        """device tables [start | lr | wd | is_matrix] per arena parameter.  The layout (start, mat)
        and the active mask are rebuilt only when the arena or the set of parameters with gradients
        changes; a per-iteration learning-rate schedule (`lr_sched.adjust_learning_rate`) only
        rewrites the small pinned host tables and refreshes the device copies with two non-blocking
        copies on the compute stream — no host synchronisation per step."""
        n = len(a.params)
        group_of = {}
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                group_of[id(p)] = gi
        active = tuple(id(p) in group_of and p.requires_grad and p.grad is not None for p in a.params)
        hyper = tuple((g["lr"], g["weight_decay"]) for g in self.param_groups)
        layout_key = (id(a), active)
        if layout_key != self._tables_key:
            dev = a.data.device
            pin = dev.type == "cuda"
            start = torch.tensor(list(a.offsets) + [a.numel], dtype=torch.int64)
            mat = torch.tensor([1.0 if (active[i] and p.ndim > 1) else 0.0 for i, p in enumerate(a.params)])
            self._host = (torch.empty(n, pin_memory=pin), torch.empty(n, pin_memory=pin))
            self._dev = [start.to(dev), torch.empty(n, device=dev), torch.empty(n, device=dev), mat.to(dev)]
            self._gidx = [group_of.get(id(p), -1) if active[i] else -1 for i, p in enumerate(a.params)]
            self._tables_key, self._hyper = layout_key, None
        if hyper != self._hyper:
            lr_h, wd_h = self._host
            lrs = [hyper[gi][0] if gi >= 0 else -1.0 for gi in self._gidx]
            wds = [hyper[gi][1] if gi >= 0 else 0.0 for gi in self._gidx]
            lr_h.copy_(torch.tensor(lrs))
            wd_h.copy_(torch.tensor(wds))
            self._dev[1].copy_(lr_h, non_blocking=True)
            self._dev[2].copy_(wd_h, non_blocking=True)
            self._hyper = hyper
        return tuple(self._dev) + (n,)

    # ------------------------------------------------------------------ checkpointing
    def _state_buffers(self):
        raise NotImplementedError

    def state_dict(self):
        """moments / momentum as flat arena-shaped fp32 tensors + step count + param_groups (without
        the parameter objects) — what `save_model` (`Models/mae/util/misc.py:301-307`) and
        `main_moco.py:310-316` store under "optimizer" for `--resume`"""
        groups = [{k: v for k, v in g.items() if k != "params"} | {"n_params": len(g["params"])}
                  for g in self.param_groups]
        return {"kind": type(self).__name__, "step_count": self.step_count, "param_groups": groups,
                "state": {k: (v.detach().clone() if v is not None else None)
                          for k, v in self._state_buffers().items()}}

    def load_state_dict(self, sd):
        assert sd["kind"] == type(self).__name__, (sd["kind"], type(self).__name__)
        assert len(sd["param_groups"]) == len(self.param_groups)
        for g, saved in zip(self.param_groups, sd["param_groups"]):
            assert saved["n_params"] == len(g["params"]), "parameter groups differ from the saved ones"
            g.update({k: v for k, v in saved.items() if k != "n_params"})
        a = self._arena()
        for k, v in sd["state"].items():
            if v is not None:
                assert v.numel() == a.numel, "optimizer state was saved for a different arena layout"
                setattr(self, k, v.to(a.data.device, torch.float32).clone())
        self.step_count = int(sd["step_count"])
        self._hyper = None

    def zero_grad(self, set_to_none: bool = True):
        for g in self.param_groups:
            for p in g["params"]:
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.zero_()


class ArenaAdamW(_ArenaOptimizer):
    """`ArenaAdamW(model, params_or_groups, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
    overlap_backward=False)`

    overlap_backward: the update of a transformer block's parameters is enqueued on a side stream as
    soon as that block's backward kernels are (engine.GradSink.tracker), so it runs behind the rest of
    the backward pass in the CUs the data-gradient chain leaves idle; `step()` then updates whatever is
    left and joins the streams.  Same arithmetic, same result.  It engages from the second step on
    (the first one learns which parameters get gradients), only while every parameter is used ONCE per
    backward pass (two-view models accumulate over uses), never with gradient accumulation over
    several backward passes, and not under parallel.DataParallel (which owns the tracker: its
    gradient all-reduce has to come first).  Measured on the MAE ViT-B step it does NOT pay (24.1 vs
    23.9-24.0 ms, same box): the HBM-bound update competes with the backward GEMMs' operand traffic for
    about what it hides.  Kept, off by default, because the result is bitwise the plain step's
    (tests/test_gpu_optim.py) and a model with more idle CUs in its backward may differ."""

    def __init__(self, model, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2,
                 overlap_backward=False):
        super().__init__(model, params, dict(lr=lr, weight_decay=weight_decay))
        self.betas, self.eps = betas, eps
        self.exp_avg = self.exp_avg_sq = None
        self._overlap = bool(overlap_backward)
        self._stream = None        # side stream of the in-backward updates
        self._done = []            # [lo, hi) ranges already updated in the current step
        self._tables = None        # device tables of the current step (built at its first use)
        self._seen = {}            # id(p) -> tracker calls in the current pass
        self._hooked = False
        self._lp = None

    def _state_buffers(self):
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq}

    # ------------------------------------------------------------------ in-backward updates
    def _hook(self):
        if self._overlap and not self._hooked:
            sink = self.model.sink()
            if sink.tracker is None:
                sink.tracker = self._on_use_done
                self._hooked = True
            else:
                self._overlap = False  # somebody else (DataParallel) owns the engine's call-back

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad(set_to_none)
        self._hook()
        self._done, self._seen, self._tables = [], {}, None

    def _launch(self, a, lo, hi, st):
        start, lr, wd, _, S = self._tables
        _lib.check(_lib.load().ssl4gie_adamw_arena_range(
            ptr(a.data), ptr(a.grad), ptr(self.exp_avg), ptr(self.exp_avg_sq), ptr(start), ptr(lr), ptr(wd), S,
            self.betas[0], self.betas[1], self.eps, self.step_count + 1, lo, hi, ptr(self._lp), st),
            "adamw_arena_range")

    @torch.no_grad()
    def _on_use_done(self, params):
        if not self._overlap or self.exp_avg is None or self._tables_key is None:
            return  # first step: plain update in step()
        for p in params:
            n = self._seen.get(id(p), 0) + 1
            self._seen[id(p)] = n
            if n > 1:  # a second use in one pass: gradients still accumulating — not this model
                self._overlap = False
                return
        a = self.model.arena()
        lo, hi = a.span(params)
        inside = [q for q, o in zip(a.params, a.offsets) if lo <= o < hi]
        ids = {id(p) for p in params}
        if any(id(q) not in ids for q in inside) or any(lo < h and l < hi for l, h in self._done):
            return  # not a contiguous run of exactly these parameters: left to step()
        if self._tables is None:
            self._lp = a.lp_flat_for_update()
            self._tables = self._build_tables_cached(a)
        if self._stream is None:
            self._stream = torch.cuda.Stream(device=a.data.device)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(a.data.device))  # the block's backward, incl. its dW join
        self._stream.wait_event(ev)
        # ... unless the block's dW products were deferred to the library's weight-gradient stream
        # (SSL4GIE_WGRAD_GROUP): wait for the latest group of either slot, as DataParallel does
        L = _lib.load()
        for slot in (0, 1):
            L.ssl4gie_wgrad_wait(slot, self._stream.cuda_stream)
        self._launch(a, lo, hi, self._stream.cuda_stream)
        self._done.append((lo, hi))

    def _build_tables_cached(self, a):
        """tables with the ACTIVE mask of the previous step (gradients do not exist yet while the
        backward pass is running); only the lr / wd values are refreshed"""
        saved = self._tables_key
        n = len(a.params)
        group_of = {id(p): gi for gi, g in enumerate(self.param_groups) for p in g["params"]}
        hyper = tuple((g["lr"], g["weight_decay"]) for g in self.param_groups)
        if hyper != self._hyper:
            lr_h, wd_h = self._host
            lr_h.copy_(torch.tensor([hyper[gi][0] if gi >= 0 else -1.0 for gi in self._gidx]))
            wd_h.copy_(torch.tensor([hyper[gi][1] if gi >= 0 else 0.0 for gi in self._gidx]))
            self._dev[1].copy_(lr_h, non_blocking=True)
            self._dev[2].copy_(wd_h, non_blocking=True)
            self._hyper = hyper
        assert self._tables_key == saved and group_of is not None
        return tuple(self._dev) + (n,)

    @torch.no_grad()
    def step(self):
        a = self._arena()
        if self.exp_avg is None or self.exp_avg.numel() != a.numel:
            self.exp_avg = torch.zeros_like(a.data)
            self.exp_avg_sq = torch.zeros_like(a.data)
        key_before = self._tables_key
        tables = self._build_tables(a)
        if self._done and self._tables_key != key_before:
            raise RuntimeError("ArenaAdamW(overlap_backward=True): the set of parameters with gradients "
                               "changed between steps after part of this step was already applied")
        self._tables = tables
        # the kernels also write the bf16 operand copy of what they update into the arena's flat shadow
        # (engine.ParamArena.lp_views): the next forward then only needs the batched transposes
        self._lp = a.lp_flat_for_update() if a.data.is_cuda else None
        st = stream()
        pos = 0
        for lo, hi in sorted(self._done) + [(a.numel, a.numel)]:  # the complement of the ranges done
            if lo > pos:
                self._launch(a, pos, lo, st)
            pos = max(pos, hi)
        if self._done:
            torch.cuda.current_stream(a.data.device).wait_stream(self._stream)
        self.step_count += 1
        self._done, self._seen, self._tables = [], {}, None
        bump_weights_epoch(touched=[p for g in self.param_groups for p in g["params"]])
        if self._lp is not None:
            a.lp_flat_is_current()


class ArenaLARS(_ArenaOptimizer):
    """`ArenaLARS(model, params_or_groups, lr=0, weight_decay=0, momentum=0.9, trust_coefficient=0.001)`"""

    def __init__(self, model, params, lr=0.0, weight_decay=0.0, momentum=0.9, trust_coefficient=0.001):
        super().__init__(model, params, dict(lr=lr, weight_decay=weight_decay))
        self.momentum, self.trust = momentum, trust_coefficient
        self.mu = None
        self._ws = None

    def _state_buffers(self):
        return {"mu": self.mu}

    @torch.no_grad()
    def step(self):
        a = self._arena()
        if self.mu is None or self.mu.numel() != a.numel:
            self.mu = torch.zeros_like(a.data)
        start, lr, wd, mat, S = self._build_tables(a)
        L = _lib.load()
        nbytes = L.ssl4gie_lars_workspace_bytes(S)
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=a.data.device)
        self.step_count += 1
        _lib.check(L.ssl4gie_lars_arena(ptr(a.data), ptr(a.grad), ptr(self.mu), ptr(start), ptr(lr), ptr(wd),
                                        ptr(mat), S, self.momentum, self.trust, ptr(self._ws), a.numel,
                                        stream()), "lars_arena")
        bump_weights_epoch(touched=[p for g in self.param_groups for p in g["params"]])
