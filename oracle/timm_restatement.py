"""CPU restatement of the timm==0.6.12 ViT building blocks the reference imports.

TEST INFRASTRUCTURE (oracle).  The reference (`/root/reference/requirements.txt:7`) pins
timm 0.6.12 but does not vendor it, and timm is not installed in this image.  These
classes restate timm's *published* algorithm for `PatchEmbed`, `Attention`, `Mlp`, `Block`
and `VisionTransformer` with identical attribute / state_dict names, so
that the reference's own `Models/mae/models_mae.py:17`, `Models/moco_v3/vits.py:13-15` and
`Models/models.py:7` can be imported on top of them when golden vectors are generated
(`tests/golden/make_golden.py`).  The attention arithmetic is additionally pinned by the
reference's own `WindowedAttention.forward` (`Models/models.py:195-209`), which is the same
formulation plus a window permutation.

Parity note: because timm itself is absent, parity at the timm boundary is pinned only by
torch op semantics + the reference's call sites; see DESIGN.md ("oracle").
"""
from __future__ import annotations

import math
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F


def to_2tuple(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


def trunc_normal_(t, std=0.02):
    # timm trunc_normal_(std=.02) with cut-offs at +-2.0 absolute: for std=.02 the cut is 100
    # sigma away, i.e. a plain normal (the reference says the same at models_mae.py:77).
    return nn.init.trunc_normal_(t, std=std, a=-2.0, b=2.0)


class PatchEmbed(nn.Module):
    """Conv2d(k=p, s=p) then flatten(2).transpose(1, 2)  (call site models_mae.py:33,152)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768,
                 norm_layer=None, flatten=True):
        super().__init__()
        self.img_size = to_2tuple(img_size)
        self.patch_size = to_2tuple(patch_size)
        self.grid_size = (self.img_size[0] // self.patch_size[0],
                          self.img_size[1] // self.patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.flatten = flatten
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size,
                              stride=self.patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

    def forward(self, x):
        _, _, h, w = x.shape
        assert h == self.img_size[0] and w == self.img_size[1], "input size mismatch"
        x = self.proj(x)
        if self.flatten:
            x = x.flatten(2).transpose(1, 2)
        return self.norm(x)


class Attention(nn.Module):
    """Same formulation as the reference's WindowedAttention (models.py:155-210) w/o windows."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        assert dim % num_heads == 0
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, 3 * dim, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x):
        b, n, c = x.shape
        qkv = self.qkv(x).reshape(b, n, 3, self.num_heads, c // self.num_heads)
        q, k, v = qkv.permute(2, 0, 3, 1, 4).unbind(0)
        a = (q @ k.transpose(-2, -1)) * self.scale
        a = self.attn_drop(a.softmax(dim=-1))
        x = (a @ v).transpose(1, 2).reshape(b, n, c)
        return self.proj_drop(self.proj(x))


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None,
                 act_layer=nn.GELU, drop=0.0):
        super().__init__()
        hidden_features = hidden_features or in_features
        out_features = out_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop2 = nn.Dropout(drop)

    def forward(self, x):
        return self.drop2(self.fc2(self.drop1(self.act(self.fc1(x)))))


class Block(nn.Module):
    """Pre-LN block: x += attn(norm1(x)); x += mlp(norm2(x)).  ls*/drop_path* are Identity at
    every reference call site (models_mae.py:40,54; models.py:277-279)."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, drop=0.0,
                 attn_drop=0.0, init_values=None, drop_path=0.0, act_layer=nn.GELU,
                 norm_layer=nn.LayerNorm):
        super().__init__()
        assert init_values is None and drop_path == 0.0, "not used by the reference"
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias,
                              attn_drop=attn_drop, proj_drop=drop)
        self.ls1 = nn.Identity()
        self.drop_path1 = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.ls2 = nn.Identity()
        self.drop_path2 = nn.Identity()

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class VisionTransformer(nn.Module):
    """The subset of timm's VisionTransformer the reference touches (models.py:262-356,
    vits.py:25-69): patch_embed, cls_token, pos_embed, pos_drop, blocks (nn.Sequential), norm,
    head, embed_dim, num_tokens, _pos_embed()."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000,
                 global_pool="token", embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0,
                 qkv_bias=True, init_values=None, class_token=True, no_embed_class=False,
                 pre_norm=False, fc_norm=None, drop_rate=0.0, attn_drop_rate=0.0,
                 drop_path_rate=0.0, weight_init="", embed_layer=PatchEmbed, norm_layer=None,
                 act_layer=None, block_fn=Block):
        super().__init__()
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)
        act_layer = act_layer or nn.GELU
        self.num_classes = num_classes
        self.global_pool = global_pool
        self.num_features = self.embed_dim = embed_dim
        self.num_tokens = 1 if class_token else 0
        self.num_prefix_tokens = self.num_tokens
        self.no_embed_class = no_embed_class
        self.patch_embed = embed_layer(img_size=img_size, patch_size=patch_size,
                                       in_chans=in_chans, embed_dim=embed_dim)
        n = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim)) if class_token else None
        self.pos_embed = nn.Parameter(torch.randn(1, n + self.num_tokens, embed_dim) * 0.02)
        self.pos_drop = nn.Dropout(p=drop_rate)
        self.norm_pre = nn.Identity()
        self.blocks = nn.Sequential(*[
            block_fn(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio,
                     qkv_bias=qkv_bias, init_values=init_values, drop=drop_rate,
                     attn_drop=attn_drop_rate, drop_path=0.0, norm_layer=norm_layer,
                     act_layer=act_layer)
            for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.fc_norm = nn.Identity()
        self.head = nn.Linear(embed_dim, num_classes) if num_classes > 0 else nn.Identity()
        # default timm init
        trunc_normal_(self.pos_embed, std=0.02)
        if self.cls_token is not None:
            nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                trunc_normal_(m.weight, std=0.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def _pos_embed(self, x):
        if self.cls_token is not None:
            x = torch.cat((self.cls_token.expand(x.shape[0], -1, -1), x), dim=1)
        return self.pos_drop(x + self.pos_embed)

    def forward_features(self, x):
        x = self._pos_embed(self.patch_embed(x))
        return self.norm(self.blocks(x))

    def forward(self, x):
        x = self.forward_features(x)
        x = x[:, 0] if self.global_pool == "token" else x[:, self.num_tokens:].mean(1)
        return self.head(x)


def _cfg(**kw):  # vits.py:13 imports it; only used to fill default_cfg
    return dict(kw)


def add_weight_decay(model, weight_decay=1e-5, skip_list=()):
    """timm.optim.optim_factory.add_weight_decay (main_pretrain.py:179): 1-D params and
    `.bias` get wd 0."""
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if p.ndim <= 1 or name.endswith(".bias") or name in skip_list:
            no_decay.append(p)
        else:
            decay.append(p)
    return [{"params": no_decay, "weight_decay": 0.0},
            {"params": decay, "weight_decay": weight_decay}]


def install_as_timm():
    """Register this module under the `timm.*` names the reference imports.  Only
    tests/golden/make_golden.py calls this (in the authoring container)."""
    import sys
    import types

    me = sys.modules[__name__]
    pkgs = {}
    for name in ("timm", "timm.models", "timm.models.layers", "timm.models.layers.helpers",
                 "timm.models.vision_transformer", "timm.models.hub", "timm.optim",
                 "timm.optim.optim_factory"):
        m = types.ModuleType(name)
        m.__path__ = []  # mark as package
        pkgs[name] = m
        sys.modules[name] = m
    pkgs["timm"].__version__ = "0.6.12"
    for n in ("PatchEmbed", "Attention", "Mlp", "Block", "VisionTransformer", "_cfg"):
        setattr(pkgs["timm.models.vision_transformer"], n, getattr(me, n))
    pkgs["timm.models.layers"].PatchEmbed = PatchEmbed
    pkgs["timm.models.layers"].trunc_normal_ = trunc_normal_
    pkgs["timm.models.layers.helpers"].to_2tuple = to_2tuple

    def _no_network(*a, **k):
        raise RuntimeError("no network in this environment")

    pkgs["timm.models.hub"].download_cached_file = _no_network
    pkgs["timm.optim.optim_factory"].add_weight_decay = add_weight_decay
    pkgs["timm.optim"].optim_factory = pkgs["timm.optim.optim_factory"]
    pkgs["timm"].models = pkgs["timm.models"]
    pkgs["timm"].optim = pkgs["timm.optim"]
    pkgs["timm.models"].layers = pkgs["timm.models.layers"]
    pkgs["timm.models"].vision_transformer = pkgs["timm.models.vision_transformer"]
    pkgs["timm.models"].hub = pkgs["timm.models.hub"]
