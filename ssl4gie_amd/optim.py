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
        self._tables = None
        self.step_count = 0

    # ------------------------------------------------------------------ segment tables
    def _arena(self):
        a = self.model.arena()
        own = [p for g in self.param_groups for p in g["params"]]
        assert all(a.owns(p) for p in own), "every optimised parameter must live in the model's arena"
        return a

    def _build_tables(self, a):
        n = len(a.params)
        group_of = {}
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                group_of[id(p)] = gi
        active = tuple(id(p) in group_of and p.requires_grad and p.grad is not None for p in a.params)
        hyper = tuple((g["lr"], g["weight_decay"]) for g in self.param_groups)
        key = (id(a), active, hyper)
        if key == self._tables_key:
            return self._tables
        dev = a.data.device
        start = torch.tensor(list(a.offsets) + [a.numel], dtype=torch.int64)
        lr = torch.full((n,), -1.0)
        wd = torch.zeros(n)
        mat = torch.zeros(n)
        for i, p in enumerate(a.params):
            if active[i]:
                g = self.param_groups[group_of[id(p)]]
                lr[i], wd[i] = g["lr"], g["weight_decay"]
                mat[i] = 1.0 if p.ndim > 1 else 0.0
        self._tables = tuple(t.to(dev) for t in (start, lr, wd, mat)) + (n,)
        self._tables_key = key
        return self._tables

    def zero_grad(self, set_to_none: bool = True):
        for g in self.param_groups:
            for p in g["params"]:
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.zero_()


class ArenaAdamW(_ArenaOptimizer):
    """`ArenaAdamW(model, params_or_groups, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)`"""

    def __init__(self, model, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(model, params, dict(lr=lr, weight_decay=weight_decay))
        self.betas, self.eps = betas, eps
        self.exp_avg = self.exp_avg_sq = None

    @torch.no_grad()
    def step(self):
        a = self._arena()
        if self.exp_avg is None or self.exp_avg.numel() != a.numel:
            self.exp_avg = torch.zeros_like(a.data)
            self.exp_avg_sq = torch.zeros_like(a.data)
        start, lr, wd, _, S = self._build_tables(a)
        self.step_count += 1
        _lib.check(_lib.load().ssl4gie_adamw_arena(ptr(a.data), ptr(a.grad), ptr(self.exp_avg),
                                                   ptr(self.exp_avg_sq), ptr(start), ptr(lr), ptr(wd), S,
                                                   self.betas[0], self.betas[1], self.eps, self.step_count,
                                                   a.numel, stream()), "adamw_arena")
        bump_weights_epoch()


class ArenaLARS(_ArenaOptimizer):
    """`ArenaLARS(model, params_or_groups, lr=0, weight_decay=0, momentum=0.9, trust_coefficient=0.001)`"""

    def __init__(self, model, params, lr=0.0, weight_decay=0.0, momentum=0.9, trust_coefficient=0.001):
        super().__init__(model, params, dict(lr=lr, weight_decay=weight_decay))
        self.momentum, self.trust = momentum, trust_coefficient
        self.mu = None
        self._ws = None

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
        bump_weights_epoch()
