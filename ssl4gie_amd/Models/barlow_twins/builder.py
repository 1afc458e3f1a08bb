"""Barlow Twins on the MI355X engine (BASELINE.json configs[4]).

The reference repository carries no Barlow Twins trainer — only a loader for ResNet50 weights
pretrained with it (`utils.py:4-5`, README pointing at the upstream project).  The specification
implemented here is therefore the BUILD'S OWN, restated from the published method (Zbontar et al.,
"Barlow Twins: Self-Supervised Learning via Redundancy Reduction", ICML 2021, Algorithm 1 and §3):

    z_A, z_B = projector(backbone(y_A)), projector(backbone(y_B))      # projector: 3 x 8192,
    c = BN(z_A)^T BN(z_B) / N_global                                   #   Linear(no bias)-BN-ReLU x2
    all_reduce(c)  (sum over ranks)                                    #   + Linear(no bias)
    loss = sum_i (1 - c_ii)^2 + lambda * sum_{i != j} c_ij^2,  lambda = 0.0051

Parity is UNPINNED by the reference (nothing to pin it to); the CPU oracle `oracle/bt_ref.py` is a
plain fp32 torch restatement of the same equations and the tests compare against it.

What runs where:
  * backbone (ViT-B trunk or ResNet50) and projector GEMMs / BatchNorms: libssl4gie_hip.so;
  * the D x D cross-correlation is ONE split-K TN GEMM over the batch (the weight-gradient kernel:
    c = zn_A^T zn_B), its gradient two NT GEMMs against bf16 copies of dL/dc and dL/dc^T;
  * the loss reduction over c and dL/dc (elementwise on a D x D fp32 matrix) stay host-side torch
    ("loss reductions" in BASELINE.json's north_star); the cross-rank exchange is one all_reduce
    of c (RCCL), exactly the exchange config 5 names.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ...engine import EngineModule
from ...resnet_engine import BatchNormFn
from ... import ops
from ..moco_v3.moco.builder import MoCo


def cross_corr_loss_terms(c, lambd):
    """loss and dL/dc of the Barlow Twins objective for a (globally reduced) correlation matrix"""
    d = torch.diagonal(c)
    on_diag = (d - 1).pow(2).sum()
    off_diag = c.pow(2).sum() - d.pow(2).sum()
    loss = on_diag + lambd * off_diag
    dc = c * (2.0 * lambd)
    torch.diagonal(dc).copy_(2.0 * (d - 1))
    return loss, dc


def exchange_cross_corr(c, group=None):
    """the one collective of the Barlow Twins step: sum of the per-rank D x D correlation blocks
    (each already divided by the GLOBAL batch size), in place.  RCCL on the GPUs, gloo in the tests."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(c, group=group)
    return c


class CrossCorrLossFn(torch.autograd.Function):
    """loss(zn1, zn2) with zn* [N_local, D] already batch-normalised (operand dtype)."""

    @staticmethod
    def forward(ctx, zn1, zn2, lambd, n_global, group):
        zn1, zn2 = zn1.contiguous(), zn2.contiguous()
        c = ops.linear_bwd_weight(zn1, zn2)  # fp32 [D, D] = zn1^T zn2 (split-K TN GEMM over the batch)
        c.div_(n_global)
        exchange_cross_corr(c, group)
        loss, dc = cross_corr_loss_terms(c, lambd)
        ctx.save_for_backward(zn1, zn2, dc)
        ctx.n_global = n_global
        return loss

    @staticmethod
    def backward(ctx, g):
        zn1, zn2, dc = ctx.saved_tensors
        dt = zn1.dtype
        dc = dc * (g / ctx.n_global)  # dL/d(zn1^T zn2); every rank's local product sees the same dL/dc
        if dt == torch.float32:
            w, wt = dc, dc.t().contiguous()
        else:
            w, wt = ops.cast(dc, dt), ops.cast_transpose(dc, dt)
        dzn1 = ops.linear_fwd(zn2, w)   # [N, D_i] = sum_j zn2[n, j] dc[i, j]
        dzn2 = ops.linear_fwd(zn1, wt)  # [N, D_j] = sum_i zn1[n, i] dc[i, j]
        return dzn1, dzn2, None, None, None


class BarlowTwins(EngineModule):
    """`BarlowTwins(backbone, feat_dim, projector="8192-8192-8192", lambd=0.0051)`;
    `forward(y1, y2) -> loss`.  `backbone` is an engine module exposing `forward_cls(imgs)`
    (ViT trunks) or `pooled(imgs)` (ResNet50) -> [N, feat_dim]."""

    def __init__(self, backbone, feat_dim, projector="8192-8192-8192", lambd=0.0051, group=None):
        super().__init__()
        self.backbone = self.adopt(backbone)
        sizes = [feat_dim] + [int(s) for s in projector.split("-")]
        layers = []
        for i in range(len(sizes) - 2):
            layers += [nn.Linear(sizes[i], sizes[i + 1], bias=False), nn.BatchNorm1d(sizes[i + 1]),
                       nn.ReLU(inplace=True)]
        layers.append(nn.Linear(sizes[-2], sizes[-1], bias=False))
        self.projector = nn.Sequential(*layers)
        self.bn = nn.BatchNorm1d(sizes[-1], affine=False)  # normalises z along the batch
        self.lambd = lambd
        self.group = group

    run_mlp = MoCo.run_mlp  # Linear(no bias) -> BatchNorm1d -> ReLU stages on the HIP engine

    def features(self, imgs):
        bb = self.backbone
        return bb.forward_cls(imgs) if hasattr(bb, "forward_cls") else bb.pooled(imgs)

    def embed(self, imgs):
        z = self.run_mlp(self.projector, self.features(imgs)).to(self.dtype_)
        return BatchNormFn.apply(z.contiguous(), None, None, None, self.bn, False, self.sink())

    def forward(self, y1, y2):
        import torch.distributed as dist
        self._prepare()
        world = 1
        if dist.is_available() and dist.is_initialized():
            world = dist.get_world_size(self.group)
        zn1, zn2 = self.embed(y1), self.embed(y2)
        return CrossCorrLossFn.apply(zn1, zn2, self.lambd, y1.shape[0] * world, self.group)
