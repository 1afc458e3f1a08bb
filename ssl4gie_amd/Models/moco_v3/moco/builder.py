"""MoCo-v3 wrapper on the MI355X engine — drop-in for the reference's
`Models/moco_v3/moco/builder.py:11-137`: `MoCo(base_encoder, dim=256, mlp_dim=4096, T=1.0)` with
`forward(x1, x2, m) -> loss`, subclasses `MoCo_ResNet` / `MoCo_ViT`, state_dict keys
`base_encoder.*`, `momentum_encoder.*`, `predictor.*` (checkpoints of `main_moco.py:310-316` drop in).

What runs where:
  * both encoders and all MLP layers (Linear without bias -> BatchNorm1d -> ReLU, :36-52) run on
    libssl4gie_hip.so (GEMMs + the BatchNorm kernels; SyncBatchNorm when the model was converted and
    world_size > 1, main_moco.py:196);
  * the momentum update (:57-61) is ONE axpby kernel over the parameter arena: base and momentum
    encoders are registered back to back with identical layouts, so `zip(parameters)` is a slice pair;
  * the InfoNCE loss (:63-73: normalise, all_gather of the keys, [N, N*W] logits, cross entropy x 2T)
    stays host-side torch on [N, 256] tensors — "loss reductions" in BASELINE.json's north_star.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ....engine import EngineModule, LinearFn, MatmulNTFn, bump_weights_epoch
from ....resnet_engine import BatchNormFn
from .... import ops


class MoCo(EngineModule):
    def __init__(self, base_encoder, dim=256, mlp_dim=4096, T=1.0):
        super().__init__()
        self.T = T
        self.base_encoder = self.adopt(base_encoder(num_classes=mlp_dim))
        self.momentum_encoder = self.adopt(base_encoder(num_classes=mlp_dim))
        self._build_projector_and_predictor_mlps(dim, mlp_dim)
        for pb, pm in zip(self.base_encoder.parameters(), self.momentum_encoder.parameters()):
            pm.data.copy_(pb.data)     # initialise
            pm.requires_grad = False   # not updated by gradient

    def _build_mlp(self, num_layers, input_dim, mlp_dim, output_dim, last_bn=True):
        mlp = []
        for l in range(num_layers):
            d1 = input_dim if l == 0 else mlp_dim
            d2 = output_dim if l == num_layers - 1 else mlp_dim
            mlp.append(nn.Linear(d1, d2, bias=False))
            if l < num_layers - 1:
                mlp.append(nn.BatchNorm1d(d2))
                mlp.append(nn.ReLU(inplace=True))
            elif last_bn:
                mlp.append(nn.BatchNorm1d(d2, affine=False))  # SimCLR design, gamma removed
        return nn.Sequential(*mlp)

    def _build_projector_and_predictor_mlps(self, dim, mlp_dim):
        pass

    # ------------------------------------------------------------------ engine pieces
    def run_mlp(self, mlp: nn.Sequential, x):
        """x [N, C] (any float dtype) -> fp32 [N, out]: Linear(no bias) + BN1d (+ ReLU) per stage"""
        h = x.to(self.dtype_).contiguous()
        mods = list(mlp)
        i = 0
        while i < len(mods):
            lin = mods[i]
            assert isinstance(lin, nn.Linear) and lin.bias is None
            h = LinearFn.apply(h, lin.weight, None, self.dtype_, self.dtype_, self.sink(), self.lp_cache)
            i += 1
            if i < len(mods) and isinstance(mods[i], (nn.BatchNorm1d, nn.SyncBatchNorm)):
                bn = mods[i]
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                h = BatchNormFn.apply(h, bn.weight, bn.bias, None, bn, relu, self.sink())
                i += 2 if relu else 1
        return h.float()

    @torch.no_grad()
    def _update_momentum_encoder(self, m):
        """param_m = param_m * m + param_b * (1 - m) over zip(parameters) (reference :57-61)"""
        a = self.arena()
        pb, pm = list(self.base_encoder.parameters()), list(self.momentum_encoder.parameters())
        b0, b1 = a.span(pb)
        m0, m1 = a.span(pm)
        assert b1 - b0 == m1 - m0, "base / momentum encoders must have identical layouts"
        ops.ema_update(a.data[m0:m1], a.data[b0:b1], m)
        # the kernel writes through raw pointers: torch's version counters do not move, so the
        # operand caches (bf16 weight copies) are invalidated through the engine's epoch instead
        bump_weights_epoch(touched=pm)

    def contrastive_loss(self, q, k):
        import torch.distributed as dist
        q = nn.functional.normalize(q, dim=1)
        k = nn.functional.normalize(k, dim=1)
        rank, world = 0, 1
        if dist.is_available() and dist.is_initialized():
            rank, world = dist.get_rank(), dist.get_world_size()
        if world > 1:
            k = concat_all_gather(k)
        logits = (MatmulNTFn.apply(q, k) if q.is_cuda else q @ k.t()) / self.T  # einsum('nc,mc->nm') of builder.py:83
        n = logits.shape[0]  # batch size per GPU
        labels = torch.arange(n, dtype=torch.long, device=logits.device) + n * rank
        return nn.functional.cross_entropy(logits, labels) * (2 * self.T)

    def encode(self, enc, x):
        raise NotImplementedError

    def forward(self, x1, x2, m):
        self._prepare()
        if self._overlap_momentum(x1):
            return self._forward_overlapped(x1, x2, m)
        q1 = self.run_mlp(self.predictor, self.encode(self.base_encoder, x1))
        q2 = self.run_mlp(self.predictor, self.encode(self.base_encoder, x2))
        with torch.no_grad():
            self._update_momentum_encoder(m)
            k1 = self.encode(self.momentum_encoder, x1)
            k2 = self.encode(self.momentum_encoder, x2)
        return self.contrastive_loss(q1, k2) + self.contrastive_loss(q2, k1)

    # ------------------------------------------------------------------ momentum branch beside the base branch
    def _overlap_momentum(self, x):
        """The momentum encoder's two forward passes (no gradient, they only feed the keys of the loss) run on
        a second stream beside the base encoder's: the small layers of one branch fill the CUs and the HBM
        gaps the other leaves (MoCo-R50 step 70.0 -> 66.7 ms, profiles/r03q_moco_overlap_ab.log).  The momentum
        update reads the base weights, which the forward does not change, so doing it FIRST is the same
        arithmetic as the reference's order (builder.py:75-96) — losses, gradients, momentum weights and
        running statistics are bit-identical (tests/test_gpu_moco.py).  SSL4GIE_MOCO_OVERLAP=0 turns it off.
        Across ranks (round 6): the momentum encoder's SyncBatchNorm layers exchange their statistics on a process
        group OF THEIR OWN (`_momentum_group`, created collectively at the first forward; the direct exchange gets a
        handle of its own through it): per communicator the collectives are issued in the same order on every rank
        — the momentum branch's in layer order, then the base branch's — whatever the two streams do on the device.
        SSL4GIE_MOCO_OVERLAP_RANKS=0 keeps the rounds 3-5 behaviour (no overlap with world_size > 1)."""
        import os
        import torch.distributed as dist
        if os.environ.get("SSL4GIE_MOCO_OVERLAP", "1") == "0" or not x.is_cuda:
            return False
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return True
        if os.environ.get("SSL4GIE_MOCO_OVERLAP_RANKS", "1") == "0":
            return False
        return self._momentum_group() is not None

    def _momentum_group(self):
        """the communicator of the momentum branch's SyncBatchNorm exchanges (None: a layer already has a group the
        caller chose — left alone, no overlap).  new_group is collective: every rank gets here in its first forward."""
        import torch.distributed as dist
        if getattr(self, "_mom_pg", None) is None:
            sbn = [m_ for m_ in self.momentum_encoder.modules() if isinstance(m_, nn.SyncBatchNorm)]
            if any(m_.process_group is not None for m_ in sbn):
                self._mom_pg = False
            else:
                self._mom_pg = dist.new_group() if sbn else True   # (no SyncBatchNorm: nothing to separate)
                if sbn:
                    for m_ in sbn:
                        m_.process_group = self._mom_pg
        return self._mom_pg or None

    def _forward_overlapped(self, x1, x2, m):
        main = torch.cuda.current_stream()
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream()
        with torch.no_grad():
            self._update_momentum_encoder(m)
            self.arena().lp_flat_for_update()      # operand copies refreshed HERE, on the main stream
            # per-batch operands both encoders share (the packed stem image) are built here as well
            warm = getattr(self.base_encoder, "prepare_inputs", None)
            if warm is not None:
                warm(x1)
                warm(x2)
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side), torch.no_grad():
            k1 = self.encode(self.momentum_encoder, x1)
            k2 = self.encode(self.momentum_encoder, x2)
            k1.record_stream(main)
            k2.record_stream(main)
        q1 = self.run_mlp(self.predictor, self.encode(self.base_encoder, x1))
        q2 = self.run_mlp(self.predictor, self.encode(self.base_encoder, x2))
        main.wait_stream(self._side)
        return self.contrastive_loss(q1, k2) + self.contrastive_loss(q2, k1)


class MoCo_ResNet(MoCo):
    def _build_projector_and_predictor_mlps(self, dim, mlp_dim):
        hidden_dim = self.base_encoder.fc.weight.shape[1]
        del self.base_encoder.fc, self.momentum_encoder.fc  # remove the original fc layer
        self.base_encoder.fc = self._build_mlp(2, hidden_dim, mlp_dim, dim)
        self.momentum_encoder.fc = self._build_mlp(2, hidden_dim, mlp_dim, dim)
        self.predictor = self._build_mlp(2, dim, mlp_dim, dim, False)

    def encode(self, enc, x):
        return self.run_mlp(enc.fc, enc.pooled(x))


class MoCo_ViT(MoCo):
    def _build_projector_and_predictor_mlps(self, dim, mlp_dim):
        hidden_dim = self.base_encoder.head.weight.shape[1]
        del self.base_encoder.head, self.momentum_encoder.head
        self.base_encoder.head = self._build_mlp(3, hidden_dim, mlp_dim, dim)
        self.momentum_encoder.head = self._build_mlp(3, hidden_dim, mlp_dim, dim)
        self.predictor = self._build_mlp(2, dim, mlp_dim, dim)

    def encode(self, enc, x):
        return self.run_mlp(enc.head, enc.forward_cls(x))


@torch.no_grad()
def concat_all_gather(tensor):
    """all_gather without gradient (reference :126-137)"""
    import torch.distributed as dist
    out = [torch.ones_like(tensor) for _ in range(dist.get_world_size())]
    dist.all_gather(out, tensor, async_op=False)
    return torch.cat(out, dim=0)
