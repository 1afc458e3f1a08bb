"""LARS as the reference's MoCo-v3 driver uses it (`Models/moco_v3/moco/optimizer.py:10-43`; an
identical copy serves MAE's linear probe, `Models/mae/util/lars.py`).  Host-side by design
(BASELINE.json north_star: the optimizer step stays Python on PyTorch-ROCm).

Update rule, per parameter with a gradient:
    if p.ndim > 1:   dp = grad + weight_decay * p
                     q  = trust_coefficient * |p| / |dp|   (1 where either norm is 0)
                     dp = dp * q
    mu = momentum * mu + dp ;  p = p - lr * mu
(no rate scaling and no weight decay for biases / norm parameters).  Written over torch._foreach
lists so that a step is a handful of multi-tensor launches instead of ~10 launches per parameter.
"""
from __future__ import annotations

import torch


class LARS(torch.optim.Optimizer):
    def __init__(self, params, lr=0, weight_decay=0, momentum=0.9, trust_coefficient=0.001):
        super().__init__(params, dict(lr=lr, weight_decay=weight_decay, momentum=momentum,
                                      trust_coefficient=trust_coefficient))

    @torch.no_grad()
    def step(self):
        for group in self.param_groups:
            matrices, vectors = [], []
            for p in group["params"]:
                if p.grad is None:
                    continue
                (matrices if p.ndim > 1 else vectors).append(p)
            updates = {}
            if matrices:
                dps = torch._foreach_add([p.grad for p in matrices], matrices, alpha=group["weight_decay"])
                pn = torch.stack(torch._foreach_norm(matrices))
                un = torch.stack(torch._foreach_norm(dps))
                q = torch.where((pn > 0) & (un > 0), group["trust_coefficient"] * pn / un.clamp(min=1e-38),
                                torch.ones_like(pn))
                torch._foreach_mul_(dps, list(q.unbind()))
                updates.update({id(p): d for p, d in zip(matrices, dps)})
            for p in vectors:
                updates[id(p)] = p.grad
            ps = matrices + vectors
            mus = []
            for p in ps:
                st = self.state[p]
                if "mu" not in st:
                    st["mu"] = torch.zeros_like(p)
                mus.append(st["mu"])
            torch._foreach_mul_(mus, group["momentum"])
            torch._foreach_add_(mus, [updates[id(p)] for p in ps])
            torch._foreach_add_(ps, mus, alpha=-group["lr"])
