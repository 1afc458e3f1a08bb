"""Parameter holders with the attribute / state_dict names of timm 0.6.12's ViT layers.

The reference builds its models from `timm.models.vision_transformer.{PatchEmbed, Block}`
(Models/mae/models_mae.py:17,33,39-41) and checkpoints are keyed by those attribute names
(`blocks.3.attn.qkv.weight`, `patch_embed.proj.bias`, ...).  These classes keep the
names and shapes — so checkpoints drop in — but hold no arithmetic: stacks of `Block`s are executed
by the native block executor (engine.run_blocks), `PatchEmbed` by engine.PatchEmbedFn.
"""
from __future__ import annotations

import torch.nn as nn


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = _pair(img_size)
        self.patch_size = _pair(patch_size)
        self.grid_size = (self.img_size[0] // self.patch_size[0],
                          self.img_size[1] // self.patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        # container for `proj.weight` [D, C, p, p] / `proj.bias` [D]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size,
                              stride=self.patch_size)
        self.norm = nn.Identity()


class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=True):
        super().__init__()
        assert dim % num_heads == 0, "dim should be divisible by num_heads"
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, in_features)


class Block(nn.Module):
    """Pre-LN block container: norm1, attn.{qkv,proj}, norm2, mlp.{fc1,fc2}."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=True, norm_layer=nn.LayerNorm):
        super().__init__()
        if not qkv_bias:
            raise NotImplementedError("the reference always passes qkv_bias=True")
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.num_heads = num_heads
