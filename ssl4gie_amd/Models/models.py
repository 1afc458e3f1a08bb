"""Finetune model zoo on the MI355X engine — mirrors the constructor API of the reference's
`Models/models.py` for the ViT-B backbones:

    ViT_from_MAE(weight_path, head, num_classes, frozen, dense, det, fixed_size, embed_dim, depth,
                 num_heads, out_token)                                  reference :360-475
    VisionTransformer_from_Any(head, num_classes, frozen, dense, det, fixed_size, embed_dim, depth,
                 num_heads, out_token, ImageNet_weights=False)          reference :262-356
    ViT_from_MoCoV3(weight_path, head, num_classes, frozen, dense, det, fixed_size, embed_dim,
                 out_token)                                             reference :478-578

state_dict keys are the reference's (`cls_token, pos_embed, patch_embed.proj.*, blocks.i.*, norm.*,
lin_head.*`; ViT_from_MAE keeps `decoder_pos_embed`, reference :395-399).  The un-masked trunk
(patch-embed GEMM, cls/pos assembly, 12 blocks with taps after blocks 2/5/8/11, final LayerNorm,
linear head) runs on libssl4gie_hip.so.

Scope (SURVEY §8, rows a8-a12 + §8f ranks 1-2): `dense=None`, `dense="depth"` and `dense="seg"` (DPT
decoder on the four tap tensors, `Models/DPT_decoder.py`) and `det=True` (windowed attention by
token order + `ViTDet_FPN`, reference :155-259) — all pinned against the reference's own classes by
tests/golden/g10_det.npz, g11_vit_api.npz, g12_resnet_dec.npz.
"""
from __future__ import annotations

import math
from functools import partial

import torch
import torch.nn as nn

from collections import OrderedDict

from .. import checkpoints, ops
from ..engine import EngineModule, LinearFn, PatchEmbedFn
from .DPT_decoder import DPT_decoder
from .resnet import ResNet50
from .mae.util.pos_embed import get_2d_sincos_pos_embed
from .vit_layers import Block, PatchEmbed

TAP_BLOCKS = (2, 5, 8, 11)  # reference models.py:453


class _ViTBackbone(EngineModule):
    """Shared trunk: patch-embed -> [cls | tokens] + pos -> blocks (-> taps) -> norm -> readout."""

    def _build_trunk(self, embed_dim, depth, num_heads, img_size=224, patch_size=16, in_chans=3,
                     mlp_ratio=4.0):
        norm_layer = partial(nn.LayerNorm, eps=1e-6)
        self.embed_dim = embed_dim
        self.num_heads = num_heads
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        n = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, embed_dim))
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias=True,
                                           norm_layer=norm_layer) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)

    def _finish(self, head, num_classes, frozen, dense, det, fixed_size, out_token):
        if det:  # reference models.py:281-285, 302-307 (and :409-414, :522-527)
            assert fixed_size % (16 * WINDOW) == 0, "fixed_size must be a multiple of 256"
            self.fixed_size = fixed_size
            self.fpn = self.adopt(ViTDet_FPN(grid=fixed_size // 16, dim=self.embed_dim))
            self.out_channels = 256
            self.patch_embed.img_size = (fixed_size, fixed_size)
            del self.cls_token
        self.head_flag = head
        if head:
            self.lin_head = nn.Linear(self.embed_dim, num_classes)
        self.frozen = frozen
        self.dense = dense
        self.det = det
        if dense:  # reference models.py:301,408,521
            self.decoder = self.adopt(DPT_decoder(num_classes=num_classes, dense=dense))
        self.out_token = out_token

    def _trunk(self, imgs, dense):
        self._prepare()
        p = self.patch_embed.patch_size[0]
        assert imgs.shape[2:] == tuple(self.patch_embed.img_size), "input size mismatch"
        tok = PatchEmbedFn.apply(imgs.float(), self.patch_embed.proj.weight,
                                 self.patch_embed.proj.bias, self.cls_token, self.pos_embed, None,
                                 self.patch_embed.num_patches, p, self.dtype_, self.sink(), self._lp)
        taps = TAP_BLOCKS if dense else ()
        tok, tap_out = self._blocks(self.blocks, tok, self.num_heads, self.norm.eps, taps=taps)
        if dense:
            return tap_out  # fp32 [B, 1+L, D] each; final norm skipped (reference :456)
        return self._ln(tok, self.norm, out_dtype=torch.float32)

    # ------------------------------------------------------------------ detection trunk
    def _pos_embed_interp(self, pos=None):
        """reference :310-320: the 14 x 14 grid part of pos_embed resized bilinearly
        (align_corners=True) to the fixed_size / 16 grid -> [1, g*g, D] (torch ops on a
        768 x 14 x 14 table: host-side plumbing, not hot-path arithmetic)"""
        pos = self.pos_embed if pos is None else pos
        return _interp_pos(pos, self.fixed_size // 16, self.embed_dim)

    def _trunk_det(self, imgs):
        """reference forward_features with det=True (:325-338).  The 16 x 16-token windows of
        WindowedAttention (:155-210) are realised by ORDER: tokens are embedded directly in
        window-major order (`perm`, the reference's own index construction), so a windowed block is
        the plain block executor on [B * windows, 256, D] and a global block the same executor on
        [B, N, D] (attention is permutation-equivariant, LayerNorm / MLP are per token); the inverse
        permutation is applied once, after the final norm."""
        from ..engine import PatchEmbedDetFn
        self._prepare()
        assert imgs.shape[2:] == (self.fixed_size, self.fixed_size), "input size mismatch"
        g = self.fixed_size // 16
        B, N, D = imgs.shape[0], g * g, self.embed_dim
        perm, inv = window_permutation(g, WINDOW, imgs.device)
        y = PatchEmbedDetFn.apply(imgs.float(), self.patch_embed.proj.weight, self.patch_embed.proj.bias,
                                  perm.unsqueeze(0).repeat(B, 1), 16, self.dtype_, self.sink(), self._lp)
        x = PosEmbedInterpAddFn.apply(y.view(B, N, D), self.pos_embed, perm, g, D, self.sink())
        nw = N // (WINDOW * WINDOW)
        i = 0
        while i < len(self.blocks):
            windowed = i in WINDOWED_BLOCKS
            j = i
            while j < len(self.blocks) and (j in WINDOWED_BLOCKS) == windowed:
                j += 1
            shape = (B * nw, WINDOW * WINDOW, D) if windowed else (B, N, D)
            x, _ = self._blocks(self.blocks[i:j], x.reshape(shape).contiguous(), self.num_heads,
                                self.norm.eps)
            i = j
        x = self._ln(x.reshape(B, N, D), self.norm, out_dtype=torch.float32)
        return x[:, inv]

    def _readout(self, x):
        if self.out_token == "cls":
            x = x[:, 0]
        elif self.out_token == "spatial":
            x = x[:, 1:].mean(1)
        if self.head_flag:
            x = self._linear(x.to(self.dtype_).contiguous(), self.lin_head, out_dtype=torch.float32)
        return x

    def forward_features(self, x, dense=None):
        if self.det:
            return self._trunk_det(x)
        return self._trunk(x, self.dense if dense is None else dense)

    def forward(self, imgs):
        if self.frozen:
            with torch.no_grad():
                x = self.forward_features(imgs)
        else:
            x = self.forward_features(imgs)
        if self.dense:  # reference models.py:346-347, 464-465, 566-567
            return self.decoder(x)
        if self.det:  # :355-356
            return self.fpn(x)
        return self._readout(x)


_INTERP_MAT = {}


def _interp_matrix(g, device, n=14):
    """[g, n] matrix of the 1-D bilinear resize n -> g with align_corners=True: row y holds the two
    weights 1 - w, w at columns i0, i0 + 1, src = y (n - 1) / (g - 1), i0 = floor(src), w = src - i0
    (what F.interpolate computes per output element; ATen's area_pixel_compute_source_index)."""
    key = (g, n, str(device))
    if key not in _INTERP_MAT:
        import numpy as np
        A = torch.zeros(g, n, dtype=torch.float32)
        scale = np.float32(n - 1) / np.float32(g - 1) if g > 1 else np.float32(0)  # ATen works in fp32
        for y in range(g):
            src = np.float32(y) * scale
            i0 = min(int(src), n - 1)
            i1 = min(i0 + 1, n - 1)
            w = np.float32(src - np.float32(i0))
            A[y, i0] += float(np.float32(1) - w)
            A[y, i1] += float(w)
        _INTERP_MAT[key] = A.to(device)
    return _INTERP_MAT[key]


def _interp_pos(pos, g, D):
    """reference `_pos_embed_interp` (models.py:310-323): the 14 x 14 grid of the position table resized
    to g x g, bilinear with align_corners=True -> [1, g*g, D].  The resize is separable and linear,
    out = (A (x) A) table: two small fp32 matmuls instead of F.interpolate, whose kernel for this tiny
    NCHW tensor takes 1.1 ms forward on the MI355X (and as long backward)."""
    A = _interp_matrix(g, pos.device)
    t = pos[0, 1:, :].reshape(14, 14 * D)                       # [i, (j, d)]
    t = (A @ t).reshape(g, 14, D)                               # rows resized: [y, j, d]
    t = torch.matmul(A, t)                                      # columns: [y, x, d] (batched over y)
    return t.reshape(1, g * g, D)


class PosEmbedInterpAddFn(torch.autograd.Function):
    """x = y + interp(pos_embed)[:, perm] for the detection trunk (reference :310-323).  An engine
    node rather than loose torch ops so that the table's gradient is written through the model's
    GradSink into the gradient ARENA like every other parameter gradient (data-parallel buckets and
    the arena optimizers read the arena, not `p.grad` objects autograd allocates elsewhere)."""

    @staticmethod
    def forward(ctx, y, pos, perm, g, D, sink):
        with torch.no_grad():
            table = _interp_pos(pos, g, D)[:, perm]
        ctx.save_for_backward(pos, perm)
        ctx.cfg = (g, D, sink)
        return y + table

    @staticmethod
    def backward(ctx, dx):
        pos, perm = ctx.saved_tensors
        g, D, sink = ctx.cfg
        (tp,), acc, rets = sink.plan([pos])
        if tp is not None:
            drow = torch.empty(g * g, D, dtype=dx.dtype, device=dx.device)
            drow[perm] = dx.sum(0)  # back to row-major grid order
            with torch.enable_grad():  # adjoint of the bilinear resize, by autograd on the small table
                leaf = pos.detach().requires_grad_(True)
                (gp,) = torch.autograd.grad(_interp_pos(leaf, g, D), leaf, drow[None])
            if acc:
                tp.add_(gp)
            else:
                tp.copy_(gp)
        return dx, rets[0], None, None, None, None


WINDOW = 16                                   # WindowedAttention(window_size=16), models.py:163
WINDOWED_BLOCKS = (0, 1, 3, 4, 6, 7, 9, 10)   # models.py:282
_PERM_CACHE = {}


def window_permutation(s, window=WINDOW, device="cpu"):
    """perm / inv_perm of WindowedAttention.forward (models.py:179-191): token indices of the s x s
    grid gathered window by window (row-major windows, row-major inside a window)."""
    key = (s, window, str(device))
    if key not in _PERM_CACHE:
        idxs = torch.arange(s * s).reshape(s, s)
        perm = torch.cat([idxs[i:i + window, j:j + window].reshape(-1)
                          for i in range(0, s, window) for j in range(0, s, window)])
        _PERM_CACHE[key] = (perm.to(device), torch.argsort(perm).to(device))
    return _PERM_CACHE[key]


class ViTDet_FPN(EngineModule):
    """Reference `ViTDet_FPN` (models.py:213-259) on the engine: same Sequential layout, hence the
    same state_dict keys (`fpn1.1.weight`, `fpn4.5.bias`, ...).  `grid` is the token grid of the
    backbone (64 for the reference's hard-coded 1024 x 1024 input: LayerNorm shapes (256,32,32) ...
    (256,256,256)); smaller grids scale every LayerNorm shape proportionally (used by the tests).
    Output: OrderedDict {"0": l4, "1": l3, "2": l2, "3": l1, "pool"} of fp32 NCHW maps."""

    def __init__(self, grid=64, dim=768, out=256):
        super().__init__()
        g = grid
        self.fpn1 = nn.Sequential(nn.Identity(), nn.Conv2d(dim, out, 1), nn.LayerNorm((out, g // 2, g // 2)),
                                  nn.Conv2d(out, out, 3, padding=1), nn.LayerNorm((out, g // 2, g // 2)))
        self.fpn2 = nn.Sequential(nn.Conv2d(dim, out, 1), nn.LayerNorm((out, g, g)),
                                  nn.Conv2d(out, out, 3, padding=1), nn.LayerNorm((out, g, g)))
        self.fpn3 = nn.Sequential(nn.ConvTranspose2d(dim, dim, 2, 2), nn.Conv2d(dim, out, 1),
                                  nn.LayerNorm((out, 2 * g, 2 * g)), nn.Conv2d(out, out, 3, padding=1),
                                  nn.LayerNorm((out, 2 * g, 2 * g)))
        self.fpn4 = nn.Sequential(nn.ConvTranspose2d(dim, dim, 2, 2), nn.LayerNorm((dim, 2 * g, 2 * g)),
                                  nn.Identity(), nn.ConvTranspose2d(dim, dim, 2, 2), nn.Conv2d(dim, out, 1),
                                  nn.LayerNorm((out, 4 * g, 4 * g)), nn.Conv2d(out, out, 3, padding=1),
                                  nn.LayerNorm((out, 4 * g, 4 * g)))
        self.grid = g
        # The (C, H, W) affine tables of the map LayerNorms live channels-LAST in memory — logical shape and
        # state_dict entry unchanged, strides (1, W C, C) — which is the order the channels-last kernels read and
        # write them in: no permuted copy of a 67 MB table per use, none of its gradient (engine.ParamArena keeps
        # the layout when it re-homes the parameter; without it MapLayerNormFn falls back to the copies).
        for mod in self.modules():
            if isinstance(mod, nn.LayerNorm) and len(mod.normalized_shape) == 3:
                for q in (mod.weight, mod.bias):
                    q.data = q.data.permute(1, 2, 0).contiguous().permute(2, 0, 1)
        # state_dict() returns CONTIGUOUS COPIES of these tables (safetensors.save_file and `.view(-1)`
        # reject the channels-last strides of an alias); every other entry still aliases its parameter.  Loading goes
        # through load_state_dict / load_my_state_dict, i.e. copy_ into the parameter, which keeps the layout; the
        # one idiom that no longer reaches these 18 tables is `model.state_dict()[k].copy_(v)` (INTEGRATION.md,
        # "state_dict layout").
        self._register_state_dict_hook(ViTDet_FPN._contiguous_tables)

    @staticmethod
    def _contiguous_tables(module, state_dict, prefix, local_metadata):
        for k in list(state_dict):
            v = state_dict[k]
            if k.startswith(prefix) and torch.is_tensor(v) and not v.is_contiguous():
                state_dict[k] = v.contiguous()
        return state_dict

    # ------------------------------------------------------------------ building blocks
    def _c1(self, x, conv):
        B, H, W, C = x.shape
        y = LinearFn.apply(x.reshape(-1, C), conv.weight, conv.bias, self.dtype_, self.dtype_,
                           self.sink(), self.lp_cache)
        return y.view(B, H, W, -1)

    def _c3(self, x, conv):
        from ..dpt_engine import Conv3x3Fn
        return Conv3x3Fn.apply(x, conv.weight, conv.bias, 1, False, self.sink(), self.lp_cache)

    def _ct(self, x, ct):
        from ..dpt_engine import ConvTransposeFn
        B, H, W, C = x.shape
        return ConvTransposeFn.apply(x.reshape(-1, C), ct.weight, ct.bias, B, H, W, self.sink(),
                                     self.lp_cache)

    def _mln(self, x, ln):
        from ..det_engine import MapLayerNormFn
        return MapLayerNormFn.apply(x, ln.weight, ln.bias, ln.eps, self.sink(), self.lp_cache)

    def forward(self, x):
        """x: fp32 tokens [B, g*g, C] in row-major grid order"""
        from ..det_engine import GeluMapFn, MaxPool2Fn
        self._prepare()
        B, N, C = x.shape
        g = int(N ** 0.5)
        m = x.to(self.dtype_).reshape(B, g, g, C)  # token-major rows ARE the channels-last map
        f1, f2, f3, f4 = self.fpn1, self.fpn2, self.fpn3, self.fpn4
        l1 = self._mln(self._c3(self._mln(self._c1(MaxPool2Fn.apply(m), f1[1]), f1[2]), f1[3]), f1[4])
        l2 = self._mln(self._c3(self._mln(self._c1(m, f2[0]), f2[1]), f2[2]), f2[3])
        l3 = self._mln(self._c3(self._mln(self._c1(self._ct(m, f3[0]), f3[1]), f3[2]), f3[3]), f3[4])
        h = GeluMapFn.apply(self._mln(self._ct(m, f4[0]), f4[1]))
        l4 = self._mln(self._c3(self._mln(self._c1(self._ct(h, f4[3]), f4[4]), f4[5]), f4[6]), f4[7])
        nchw = lambda t: t.permute(0, 3, 1, 2).float()
        l1o = nchw(l1)
        pool = l1o[:, :, ::2, ::2]  # max_pool2d(kernel_size=1, stride=2)
        return OrderedDict([("0", nchw(l4)), ("1", nchw(l3)), ("2", nchw(l2)), ("3", l1o), ("pool", pool)])


class ViT_from_MAE(_ViTBackbone):
    def __init__(self, weight_path, head, num_classes, frozen, dense, det, fixed_size, embed_dim,
                 depth, num_heads, out_token):
        super().__init__()
        self._build_trunk(embed_dim, depth, num_heads)
        n = self.patch_embed.num_patches
        # MAE leftovers that stay in the state_dict (reference deletes the decoder modules but not
        # decoder_pos_embed, models.py:395-399) and fixed sin-cos tables
        self.pos_embed.requires_grad = False
        self.decoder_pos_embed = nn.Parameter(torch.zeros(1, n + 1, 512), requires_grad=False)
        grid = int(n ** .5)
        with torch.no_grad():
            self.pos_embed.copy_(torch.from_numpy(
                get_2d_sincos_pos_embed(embed_dim, grid, cls_token=True)).float()[None])
            self.decoder_pos_embed.copy_(torch.from_numpy(
                get_2d_sincos_pos_embed(512, grid, cls_token=True)).float()[None])
            w = self.patch_embed.proj.weight
            nn.init.xavier_uniform_(w.view(w.shape[0], -1))
            nn.init.normal_(self.cls_token, std=.02)
            for m in self.modules():
                if isinstance(m, nn.Linear):
                    nn.init.xavier_uniform_(m.weight)
                    nn.init.zeros_(m.bias)
        if weight_path is not None:
            weights = checkpoints.load_file(weight_path)["model"]  # reference :392-394
            self.load_my_state_dict(weights)
        self._finish(head, num_classes, frozen, dense, det, fixed_size, out_token)

    def load_my_state_dict(self, state_dict):
        """Copy every tensor whose name we own; ignore the rest (reference :417-425)."""
        own = dict(self.named_parameters())
        own.update(dict(self.named_buffers()))   # the tensors themselves: a state_dict entry may be a copy (ViTDet_FPN)
        own = {k: v for k, v in own.items() if k in self.state_dict()}
        hits = 0
        with torch.no_grad():
            for name, value in state_dict.items():
                if name in own:
                    own[name].copy_(value)
                    hits += 1
        print(f"Successfully loaded params for {hits} items")

    def forward_encoder(self, x):
        return self.forward_features(x)


class VisionTransformer_from_Any(_ViTBackbone):
    def __init__(self, head, num_classes, frozen, dense, det, fixed_size, embed_dim, depth,
                 num_heads, out_token, ImageNet_weights=False):
        super().__init__()
        npz = None
        if ImageNet_weights:  # reference :286-290 downloads the augreg ViT-B/16 .npz; here: a local file
            import os
            npz = os.environ.get("SSL4GIE_AUGREG_NPZ")
            if not npz or not os.path.exists(npz):
                raise RuntimeError("ImageNet augreg weights are downloaded by the reference "
                                   "(models.py:286-290); no network here — point SSL4GIE_AUGREG_NPZ at the "
                                   "B_16-i21k-300ep-...-res_224.npz file, or load a state_dict instead")
        self._build_trunk(embed_dim, depth, num_heads)
        with torch.no_grad():  # timm default init
            nn.init.trunc_normal_(self.pos_embed, std=.02, a=-2.0, b=2.0)
            nn.init.normal_(self.cls_token, std=1e-6)
            for m in self.modules():
                if isinstance(m, nn.Linear):
                    nn.init.trunc_normal_(m.weight, std=.02, a=-2.0, b=2.0)
                    nn.init.zeros_(m.bias)
        if npz is not None:
            checkpoints.load_augreg_npz(self, npz)
        self._finish(head, num_classes, frozen, dense, det, fixed_size, out_token)


class ViT_from_MoCoV3(_ViTBackbone):
    def __init__(self, weight_path, head, num_classes, frozen, dense, det, fixed_size, embed_dim,
                 out_token):
        super().__init__()
        self._build_trunk(embed_dim, 12, 12)
        self.pos_embed.requires_grad = False
        with torch.no_grad():
            self.pos_embed.copy_(moco_sincos_pos_embed(embed_dim, self.patch_embed.grid_size))
            # MoCo-v3 init (reference Models/moco_v3/vits.py:31-47)
            for name, m in self.named_modules():
                if isinstance(m, nn.Linear):
                    if "qkv" in name:
                        val = math.sqrt(6. / float(m.weight.shape[0] // 3 + m.weight.shape[1]))
                        nn.init.uniform_(m.weight, -val, val)
                    else:
                        nn.init.xavier_uniform_(m.weight)
                    nn.init.zeros_(m.bias)
            nn.init.normal_(self.cls_token, std=1e-6)
            ps = self.patch_embed.patch_size
            val = math.sqrt(6. / float(3 * ps[0] * ps[1] + embed_dim))
            nn.init.uniform_(self.patch_embed.proj.weight, -val, val)
            nn.init.zeros_(self.patch_embed.proj.bias)
        if weight_path is not None:
            self.load_state_dict(checkpoints.load_file(weight_path))  # reference :507-509 (strict)
        self._finish(head, num_classes, frozen, dense, det, fixed_size, out_token)


def moco_sincos_pos_embed(embed_dim, grid_size, temperature=10000.):
    """Reference Models/moco_v3/vits.py:53-69: fp32, channels [sin w | cos w | sin h | cos h] with
    the first grid axis slowest (ij meshgrid), zero cls row."""
    h, w = grid_size
    gw, gh = torch.meshgrid(torch.arange(w, dtype=torch.float32),
                            torch.arange(h, dtype=torch.float32), indexing="ij")
    assert embed_dim % 4 == 0
    pd = embed_dim // 4
    omega = 1. / (temperature ** (torch.arange(pd, dtype=torch.float32) / pd))
    ow = gw.flatten()[:, None] * omega[None]
    oh = gh.flatten()[:, None] * omega[None]
    emb = torch.cat([ow.sin(), ow.cos(), oh.sin(), oh.cos()], dim=1)[None]
    return torch.cat([torch.zeros(1, 1, embed_dim), emb], dim=1)


class ResNet_Dec_Block(nn.Module):
    """parameter container of the reference's decoder bottleneck (models.py:14-41): identity branch
    (1x1 conv + BN when the input is the 2C-wide concatenation) and process = 1x1 -> BN -> ReLU ->
    3x3 -> BN -> ReLU -> 1x1 -> BN; same Sequential indices, hence the same state_dict keys."""

    def __init__(self, channels, fusion=False):
        super().__init__()
        if fusion:
            self.identity = nn.Sequential(nn.Conv2d(channels * 2, channels, 1), nn.BatchNorm2d(channels))
            conv1 = nn.Conv2d(channels * 2, channels // 4, 1)
        else:
            self.identity = nn.Identity()
            conv1 = nn.Conv2d(channels, channels // 4, 1)
        self.process = nn.Sequential(conv1, nn.BatchNorm2d(channels // 4), nn.ReLU(),
                                     nn.Conv2d(channels // 4, channels // 4, 3, padding=1),
                                     nn.BatchNorm2d(channels // 4), nn.ReLU(),
                                     nn.Conv2d(channels // 4, channels, 1), nn.BatchNorm2d(channels))
        self.relu = nn.ReLU()


class ResNet_Dec_Level(nn.Module):
    """models.py:44-60: chan_reduce (1x1 conv + BN) -> bilinear x2 -> cat with the skip map -> blocks"""

    def __init__(self, channels, n_blocks):
        super().__init__()
        self.chan_reduce = nn.Sequential(nn.Conv2d(channels * 2, channels, 1), nn.BatchNorm2d(channels))
        self.up = nn.Identity()  # nn.Upsample holds no tensors
        blocks = [ResNet_Dec_Block(channels, fusion=True)]
        for _ in range(1, n_blocks):
            blocks.append(ResNet_Dec_Block(channels, fusion=False))
        self.blocks = nn.Sequential(*blocks)


class ResNet_from_Any(ResNet50):
    """Reference `Models/models.py:63-152`: torchvision ResNet50 with `fc = Identity` (no `fc.*` keys:
    the reference assigns Identity before its strict load, :77-80), optional linear head, and — when
    `dense` — the U-Net-style decoder over the four stage maps (`decoder_levels.{0-2}.*`,
    `output_conv.*`, :88-135): every convolution is a GEMM of libssl4gie_hip.so on channels-last
    maps (1x1 = Linear with bias, 3x3 = implicit GEMM), BatchNorm fuses the ReLU / residual add,
    bilinear x2 and the ReLU -> 1x1 -> Sigmoid tail are the DPT kernels; the channel concatenation
    is a torch.cat on the last (channel) axis."""

    def __init__(self, weight_path, head, num_classes, frozen, dense, ImageNet_weights=False):
        super().__init__()
        if ImageNet_weights:
            raise RuntimeError("ImageNet weights are downloaded by the reference (models.py:70-75); no "
                               "network here — load a state_dict instead")
        self.fc = nn.Identity()
        if weight_path is not None:
            self.load_state_dict(checkpoints.load_file(weight_path))  # reference :78-80 (strict)
        self.head = head
        if head:
            self.lin_head = nn.Linear(2048, num_classes)
        self.frozen = frozen
        self.dense = dense
        if dense:
            self.decoder_levels = nn.ModuleList([ResNet_Dec_Level(1024, 3), ResNet_Dec_Level(512, 3),
                                                 ResNet_Dec_Level(256, 3)])
            # indices 0, 2 (Upsample), 4 (ReLU), 6 (Sigmoid) are parameter-free
            self.output_conv = nn.Sequential(nn.Identity(), nn.Conv2d(256, 128, 3, 1, 1), nn.Identity(),
                                             nn.Conv2d(128, 32, 3, 1, 1), nn.Identity(),
                                             nn.Conv2d(32, 1, 1, 1, 0), nn.Identity())

    # ------------------------------------------------------------------ dense decoder
    def _c1b(self, x, conv):
        """Conv2d(k=1) with bias on a channels-last map"""
        B, H, W, C = x.shape
        y = LinearFn.apply(x.reshape(-1, C), conv.weight, conv.bias, self.dtype_, self.dtype_,
                           self.sink(), self.lp_cache)
        return y.view(B, H, W, -1)

    def _dec_block(self, x, blk: ResNet_Dec_Block):
        from ..dpt_engine import Conv3x3Fn
        identity = x
        if not isinstance(blk.identity, nn.Identity):
            identity = self._bn(self._c1b(x, blk.identity[0]), blk.identity[1], False)
        p = blk.process
        out = self._bn(self._c1b(x, p[0]), p[1], True)
        out = Conv3x3Fn.apply(out, p[3].weight, p[3].bias, 1, False, self.sink(), self.lp_cache)
        out = self._bn(out, p[4], True)
        out = self._c1b(out, p[6])
        return self._bn(out, p[7], True, res=identity)  # relu(bn(out) + identity)

    def _dec_level(self, lvl: ResNet_Dec_Level, x_low, x_high):
        from ..dpt_engine import Upsample2xFn
        up = Upsample2xFn.apply(self._bn(self._c1b(x_low, lvl.chan_reduce[0]), lvl.chan_reduce[1], False))
        x = torch.cat((up, x_high), dim=3)  # channels-last: torch.cat(dim=1) of the NCHW reference
        for blk in lvl.blocks:
            x = self._dec_block(x, blk)
        return x

    def decode(self, fmaps):
        """models.py:128-135 on channels-last maps -> fp32 [B, 1, H, W]"""
        from ..dpt_engine import Conv3x3Fn, DepthHeadFn, Upsample2xFn
        out = self._dec_level(self.decoder_levels[0], fmaps[-1], fmaps[-2])
        for i in range(1, len(self.decoder_levels)):
            out = self._dec_level(self.decoder_levels[i], out, fmaps[-i - 2])
        oc = self.output_conv
        h = Upsample2xFn.apply(out)
        h = Conv3x3Fn.apply(h, oc[1].weight, oc[1].bias, 1, False, self.sink(), self.lp_cache)
        h = Upsample2xFn.apply(h)
        h = Conv3x3Fn.apply(h, oc[3].weight, oc[3].bias, 1, False, self.sink(), self.lp_cache)
        return DepthHeadFn.apply(h, oc[5].weight, oc[5].bias, self.sink())

    def forward_features(self, x):
        return self.forward_maps(x, all_stages=bool(self.dense))

    def forward(self, imgs):
        if self.dense:
            if self.frozen:
                with torch.no_grad():
                    fmaps = self.forward_maps(imgs, all_stages=True)
            else:
                fmaps = self.forward_maps(imgs, all_stages=True)
            return self.decode(fmaps)
        if self.frozen:
            with torch.no_grad():
                x = self.pooled(imgs)
        else:
            x = self.pooled(imgs)
        if self.head:
            x = LinearFn.apply(x.to(self.dtype_).contiguous(), self.lin_head.weight,
                               self.lin_head.bias, self.dtype_, torch.float32, self.sink(),
                               self.lp_cache)
        return x
