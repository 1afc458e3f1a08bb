"""CPU oracle: fp32 torch restatement of the Barlow Twins head and loss.  TEST INFRASTRUCTURE.

PARITY UNPINNED: /root/reference holds no Barlow Twins trainer (only a ResNet50 weight loader,
utils.py:4-5), so there is no reference code, test or fixture to pin this to.
The equations follow the published method (Zbontar et al., ICML 2021, Algorithm 1): projector of
Linear(no bias) - BatchNorm1d - ReLU stages + a final Linear(no bias); c = BN(z1)^T BN(z2) / N
with an affine-free BatchNorm over the batch; loss = sum_i (1 - c_ii)^2 + lambda sum_{i!=j} c_ij^2.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def projector_forward(weights, bn_params, x, eps=1e-5):
    """weights: list of [out, in]; bn_params: list of (gamma, beta) for all but the last stage"""
    h = x
    for i, w in enumerate(weights):
        h = h @ w.t()
        if i < len(weights) - 1:
            g, b = bn_params[i]
            h = F.relu(F.batch_norm(h, None, None, g, b, True, 0.1, eps))
    return h


def barlow_loss(z1, z2, lambd, eps=1e-5):
    n = z1.shape[0]
    zn1 = F.batch_norm(z1, None, None, None, None, True, 0.1, eps)
    zn2 = F.batch_norm(z2, None, None, None, None, True, 0.1, eps)
    c = zn1.t() @ zn2 / n
    on = (torch.diagonal(c) - 1).pow(2).sum()
    off = c.pow(2).sum() - torch.diagonal(c).pow(2).sum()
    return on + lambd * off
