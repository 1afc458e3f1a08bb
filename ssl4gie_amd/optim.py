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
    """`ArenaAdamW(model, params_or_groups, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)`"""

    def __init__(self, model, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(model, params, dict(lr=lr, weight_decay=weight_decay))
        self.betas, self.eps = betas, eps
        self.exp_avg = self.exp_avg_sq = None

    def _state_buffers(self):
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq}

    @torch.no_grad()
    def step(self):
        a = self._arena()
        if self.exp_avg is None or self.exp_avg.numel() != a.numel:
            self.exp_avg = torch.zeros_like(a.data)
            self.exp_avg_sq = torch.zeros_like(a.data)
        start, lr, wd, _, S = self._build_tables(a)
        self.step_count += 1
        # the kernel also writes the bf16 operand copy of what it updates into the arena's flat shadow
        # (engine.ParamArena.lp_views): the next forward then only needs the batched transposes
        lp = a.lp_flat_for_update() if a.data.is_cuda else None
        _lib.check(_lib.load().ssl4gie_adamw_arena_lp(ptr(a.data), ptr(a.grad), ptr(self.exp_avg),
                                                      ptr(self.exp_avg_sq), ptr(start), ptr(lr), ptr(wd), S,
                                                      self.betas[0], self.betas[1], self.eps,
                                                      self.step_count, a.numel, ptr(lp), stream()),
                   "adamw_arena")
        bump_weights_epoch()
        if lp is not None:
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
        bump_weights_epoch()
