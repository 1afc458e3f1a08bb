"""CPU oracle: functional fp32 restatement of the detection ViT backbone.  TEST INFRASTRUCTURE.

Follows /root/reference/Models/models.py: `WindowedAttention.forward` (:178-210: the perm /
inv_perm index construction, per-window softmax(q k^T * scale) v, inverse permutation BEFORE the
output projection), the block swap of `det=True` (:281-285: blocks 0,1,3,4,6,7,9,10 windowed),
`_pos_embed_interp` (:310-323: bilinear resize of the 14 x 14 table with align_corners=True, no cls
token), `forward_features` (:325-338) and `ViTDet_FPN` (:213-259: MaxPool2d(2) / ConvTranspose2d
(2, 2) / 1x1 / LayerNorm((C,H,W)) / GELU / 3x3 stacks, `pool` = max_pool2d(kernel 1, stride 2)).
PINNED: tests/golden/g10_det.npz holds outputs and gradients of the reference's own
`VisionTransformer_from_Any(det=True)` / `ViT_from_MAE(det=True)` / `ViTDet_FPN` (models.py imported
in the authoring container on top of oracle/timm_restatement.py + oracle/torchvision_restatement.py
— timm and torchvision themselves are pinned but absent — by tests/golden/make_golden.py:g10_det);
tests/test_oracle_golden.py checks this restatement against it.  The permutation is integer work
restated index for index (checked for bijectivity and window structure in the tests).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

WINDOWED = (0, 1, 3, 4, 6, 7, 9, 10)


def window_perm(n_tokens: int, window: int = 16):
    """models.py:180-191, statement for statement"""
    s = int(n_tokens ** 0.5)
    idxs = torch.arange(n_tokens).reshape(s, s)
    perm = []
    for i in range(0, s, window):
        for j in range(0, s, window):
            perm.append(idxs[i:i + window, j:j + window].reshape(window ** 2))
    windows = len(perm)
    perm = torch.cat(perm)
    return perm, torch.argsort(perm), windows


def windowed_attention(sd, pre, x, heads, window=16):
    B, N, C = x.shape
    perm, inv_perm, windows = window_perm(N, window)
    x = x[:, perm]
    qkv = F.linear(x, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"])
    qkv = qkv.reshape(B, windows, N // windows, 3, heads, C // heads).permute(3, 0, 1, 4, 2, 5)
    q, k, v = qkv.unbind(0)
    attn = ((q @ k.transpose(-2, -1)) * (C // heads) ** -0.5).softmax(dim=-1)
    x = (attn @ v).transpose(2, 3).reshape(B, N, C)
    x = x[:, inv_perm]
    return F.linear(x, sd[pre + "proj.weight"], sd[pre + "proj.bias"])


def global_attention(sd, pre, x, heads):
    B, N, C = x.shape
    hd = C // heads
    qkv = F.linear(x, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"])
    q, k, v = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4).unbind(0)
    a = torch.softmax((q @ k.transpose(-2, -1)) * hd ** -0.5, dim=-1)
    return F.linear((a @ v).transpose(1, 2).reshape(B, N, C), sd[pre + "proj.weight"], sd[pre + "proj.bias"])


def block(sd, pre, x, heads, eps, windowed):
    d = x.shape[-1]
    h = F.layer_norm(x, (d,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], eps)
    a = windowed_attention(sd, pre + "attn.", h, heads) if windowed else global_attention(sd, pre + "attn.", h, heads)
    x = x + a
    h = F.layer_norm(x, (d,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], eps)
    u = F.linear(h, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"])
    return x + F.linear(F.gelu(u), sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])


def pos_embed_interp(sd, fixed_size, dim=768):
    g = fixed_size // 16
    p2 = sd["pos_embed"][:, 1:, :].transpose(1, 2).reshape(1, dim, 14, 14)
    p2 = F.interpolate(p2, size=(g, g), mode="bilinear", align_corners=True)
    return p2.reshape(1, dim, g * g).transpose(1, 2)


def det_trunk(sd, imgs, fixed_size, depth=12, heads=12, eps=1e-6, dim=768):
    y = F.conv2d(imgs, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=16)
    x = y.flatten(2).transpose(1, 2) + pos_embed_interp(sd, fixed_size, dim)
    for i in range(depth):
        x = block(sd, f"blocks.{i}.", x, heads, eps, i in WINDOWED)
    return F.layer_norm(x, (dim,), sd["norm.weight"], sd["norm.bias"], eps)


def _ln_map(sd, p, x, eps=1e-5):
    return F.layer_norm(x, tuple(sd[p + ".weight"].shape), sd[p + ".weight"], sd[p + ".bias"], eps)


def fpn(sd, x, pre="fpn."):
    """x: tokens [B, N, C] -> dict of NCHW maps (models.py:250-259)"""
    B, N, C = x.shape
    H = int(N ** 0.5)
    m = torch.reshape(x.transpose(1, 2), (B, C, H, H))
    c = lambda p, t, **kw: F.conv2d(t, sd[pre + p + ".weight"], sd[pre + p + ".bias"], **kw)
    ct = lambda p, t: F.conv_transpose2d(t, sd[pre + p + ".weight"], sd[pre + p + ".bias"], stride=2)
    l1 = _ln_map(sd, pre + "fpn1.4", c("fpn1.3", _ln_map(sd, pre + "fpn1.2", c("fpn1.1", F.max_pool2d(m, 2))), padding=1))
    l2 = _ln_map(sd, pre + "fpn2.3", c("fpn2.2", _ln_map(sd, pre + "fpn2.1", c("fpn2.0", m)), padding=1))
    l3 = _ln_map(sd, pre + "fpn3.4", c("fpn3.3", _ln_map(sd, pre + "fpn3.2", c("fpn3.1", ct("fpn3.0", m))), padding=1))
    h = F.gelu(_ln_map(sd, pre + "fpn4.1", ct("fpn4.0", m)))
    l4 = _ln_map(sd, pre + "fpn4.7", c("fpn4.6", _ln_map(sd, pre + "fpn4.5", c("fpn4.4", ct("fpn4.3", h))), padding=1))
    pool = F.max_pool2d(l1, kernel_size=1, stride=2, padding=0)
    return {"0": l4, "1": l3, "2": l2, "3": l1, "pool": pool}
