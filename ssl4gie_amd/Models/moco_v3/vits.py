"""MoCo-v3 ViT variants on the engine (reference `Models/moco_v3/vits.py:25-69,96-121`): a timm
VisionTransformer with a FIXED 2-D sin-cos position embedding (cls row zero), MoCo's init (uniform
qkv per third, xavier elsewhere, zero biases, cls ~ N(0, 1e-6)), optional stop-gradient on the patch
embedding, and a `head` Linear that MoCo_ViT replaces by its projector.  `vit_base(**kw)` is the
factory `main_moco.py:181-183` calls through `partial(vits.__dict__[arch], stop_grad_conv1=...)`.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from ..models import _ViTBackbone, moco_sincos_pos_embed


class VisionTransformerMoCo(_ViTBackbone):
    def __init__(self, embed_dim=768, depth=12, num_heads=12, num_classes=1000, stop_grad_conv1=False,
                 **kwargs):
        super().__init__()
        self._build_trunk(embed_dim, depth, num_heads)
        self.head = nn.Linear(embed_dim, num_classes)
        self.pos_embed.requires_grad = False
        with torch.no_grad():
            self.pos_embed.copy_(moco_sincos_pos_embed(embed_dim, self.patch_embed.grid_size))
            for name, m in self.named_modules():
                if isinstance(m, nn.Linear):
                    if "qkv" in name:  # treat the weights of Q, K, V separately
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
        if stop_grad_conv1:
            self.patch_embed.proj.weight.requires_grad = False
            self.patch_embed.proj.bias.requires_grad = False
        self.dense, self.det, self.frozen, self.out_token, self.head_flag = None, False, False, "cls", False

    def forward_cls(self, imgs):
        """final-norm cls token, fp32 [B, D] (timm global_pool='token')"""
        return self._trunk(imgs, None)[:, 0]

    def forward(self, imgs):
        from ...engine import LinearFn
        x = self.forward_cls(imgs)
        return LinearFn.apply(x.to(self.dtype_).contiguous(), self.head.weight, self.head.bias,
                              self.dtype_, torch.float32, self.sink(), self.lp_cache)


def vit_small(**kwargs):
    return VisionTransformerMoCo(embed_dim=384, depth=12, num_heads=12, **kwargs)


def vit_base(**kwargs):
    return VisionTransformerMoCo(embed_dim=768, depth=12, num_heads=12, **kwargs)
