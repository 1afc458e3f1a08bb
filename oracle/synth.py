"""Deterministic synthetic weights / inputs shared by the golden generator and the tests.

TEST INFRASTRUCTURE.  Weights are a pure function of (key name, shape, seed) so that the
112 M-parameter ViT-B state_dict never has to be stored in a fixture: the generator
(`tests/golden/make_golden.py`) loads them into the *reference* classes, the tests load the
same tensors into the build's classes, and the fixture stores a checksum of them.
"""
from __future__ import annotations

import hashlib
import zlib

import numpy as np
import torch

from .mae_ref import MAEConfig, sincos_2d


def _gen(key: str, seed: int) -> torch.Generator:
    h = zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1 & 0xFFFFFFFF)
    return torch.Generator("cpu").manual_seed(int(h))


def _block_shapes(d: int, hidden: int):
    return {
        "norm1.weight": (d,), "norm1.bias": (d,),
        "attn.qkv.weight": (3 * d, d), "attn.qkv.bias": (3 * d,),
        "attn.proj.weight": (d, d), "attn.proj.bias": (d,),
        "norm2.weight": (d,), "norm2.bias": (d,),
        "mlp.fc1.weight": (hidden, d), "mlp.fc1.bias": (hidden,),
        "mlp.fc2.weight": (d, hidden), "mlp.fc2.bias": (d,),
    }


def mae_shapes(cfg: MAEConfig) -> dict:
    """state_dict schema of the reference MaskedAutoencoderViT (254 tensors for
    ViT-B)."""
    d, dd, p, l = cfg.embed_dim, cfg.decoder_embed_dim, cfg.patch_size, cfg.num_patches
    s = {"cls_token": (1, 1, d), "pos_embed": (1, l + 1, d),
         "patch_embed.proj.weight": (d, cfg.in_chans, p, p), "patch_embed.proj.bias": (d,)}
    for i in range(cfg.depth):
        for k, v in _block_shapes(d, int(d * cfg.mlp_ratio)).items():
            s[f"blocks.{i}.{k}"] = v
    s["norm.weight"] = (d,)
    s["norm.bias"] = (d,)
    s["decoder_embed.weight"] = (dd, d)
    s["decoder_embed.bias"] = (dd,)
    s["mask_token"] = (1, 1, dd)
    s["decoder_pos_embed"] = (1, l + 1, dd)
    for i in range(cfg.decoder_depth):
        for k, v in _block_shapes(dd, int(dd * cfg.mlp_ratio)).items():
            s[f"decoder_blocks.{i}.{k}"] = v
    s["decoder_norm.weight"] = (dd,)
    s["decoder_norm.bias"] = (dd,)
    s["decoder_pred.weight"] = (p * p * cfg.in_chans, dd)
    s["decoder_pred.bias"] = (p * p * cfg.in_chans,)
    return s


def synth_tensor(key: str, shape, seed: int) -> torch.Tensor:
    g = _gen(key, seed)
    if key.endswith("pos_embed"):
        raise ValueError("pos_embed tables are not random")
    if "norm" in key and key.endswith(".weight"):
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if key.endswith(".bias"):
        return 0.02 * torch.randn(shape, generator=g)
    if key in ("cls_token", "mask_token"):
        return 0.02 * torch.randn(shape, generator=g)
    fan_out = shape[0]
    fan_in = int(np.prod(shape[1:]))
    std = (2.0 / (fan_in + fan_out)) ** 0.5
    return std * torch.randn(shape, generator=g)


def mae_state_dict(cfg: MAEConfig, seed: int = 0) -> dict:
    sd = {}
    for k, shp in mae_shapes(cfg).items():
        if k == "pos_embed":
            sd[k] = torch.from_numpy(sincos_2d(cfg.embed_dim, cfg.grid)).float()[None]
        elif k == "decoder_pos_embed":
            sd[k] = torch.from_numpy(sincos_2d(cfg.decoder_embed_dim, cfg.grid)).float()[None]
        else:
            sd[k] = synth_tensor(k, shp, seed)
    return sd


def state_dict_digest(sd: dict) -> str:
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def synth_images(b: int, cfg: MAEConfig, seed: int = 0) -> torch.Tensor:
    """N(0,1) images ~ post-Normalize statistics."""
    g = torch.Generator("cpu").manual_seed(1000 + seed)
    return torch.randn(b, cfg.in_chans, cfg.img_size, cfg.img_size, generator=g)


def synth_noise(b: int, l: int, seed: int = 0) -> np.ndarray:
    """Masking noise in [0,1) from the host generator with tied rows redrawn, so that the
    reference's non-stable argsort and the stable rule agree."""
    g = torch.Generator("cpu").manual_seed(2000 + seed)
    noise = torch.rand(b, l, generator=g).numpy()
    for r in range(b):
        while len(np.unique(noise[r])) != l:
            noise[r] = torch.rand(l, generator=g).numpy()
    return noise


def keyed_tensor(key: str, shape, seed: int) -> torch.Tensor:
    """Generic rule for the finetune zoo (ViT trunks, ViTDet pyramid, ResNet50 + decoders, heads):
    a pure function of (key, shape, seed) like `synth_tensor`, covering BatchNorm buffers, conv /
    transposed-conv kernels and whole-map LayerNorm affines."""
    shape = tuple(shape)
    g = _gen(key, seed)
    if key.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=torch.long)
    if key.endswith("running_mean"):
        return torch.zeros(shape)
    if key.endswith("running_var"):
        return torch.ones(shape)
    if key.endswith("pos_embed") or key in ("cls_token", "mask_token"):
        return 0.02 * torch.randn(shape, generator=g)
    if key.endswith(".bias"):
        return 0.05 * torch.randn(shape, generator=g)
    if len(shape) in (1, 3):  # LayerNorm / BatchNorm scale (3-D: nn.LayerNorm((C, H, W)))
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    fan_in = int(np.prod(shape[1:]))
    return torch.randn(shape, generator=g) * (1.0 / fan_in) ** 0.5


def keyed_state_dict(shapes: dict, seed: int, keep=()) -> dict:
    """`shapes`: key -> shape (e.g. {k: v.shape for k, v in module.state_dict().items()}); keys in
    `keep` are left out (fixed tables the module builds itself)."""
    return {k: keyed_tensor(k, s, seed) for k, s in shapes.items() if k not in keep}


def depth_batches(nb=4, b=2):
    """the seeded (image, depth target) batches of the G13 curve (tests/golden/make_golden.py g13_depth_curve and
    tests/test_gpu_curves.py): batch i of the rotation.  The target is a learnable function of the image (a
    16 x 16 box blur of its channel mean squashed into (0, 1)) — a pure-noise target leaves the scale-and-shift
    invariant loss flat — with ~10 % of the pixels invalid (zero), as real depth maps have."""
    from . import mae_ref
    out = []
    for i in range(nb):
        imgs = synth_images(b, mae_ref.VIT_B, seed=300 + i)
        g = torch.Generator("cpu").manual_seed(400 + i)
        gray = imgs.mean(1, keepdim=True)
        blur = torch.nn.functional.avg_pool2d(gray, 17, stride=1, padding=8, count_include_pad=False)
        tgt = torch.sigmoid(3.0 * blur / blur.std())
        tgt = torch.where(torch.rand(b, 1, 224, 224, generator=g) < 0.1, torch.zeros(()), tgt)
        out.append((imgs, tgt))
    return out


def det_batches(nb=2, size=512, dim=768):
    """the seeded (image, target tokens) batches of the G15 curve (detection trunk at size x size, B = 1): the
    target is a fixed random token map the trunk's output is regressed on"""
    out = []
    for i in range(nb):
        g = torch.Generator("cpu").manual_seed(600 + i)
        imgs = torch.randn(1, 3, size, size, generator=g)
        tgt = 0.5 * torch.randn(1, (size // 16) ** 2, dim, generator=g)
        out.append((imgs, tgt))
    return out


def moco_views(nb=4, b=8, size=64):
    """the seeded view pairs of the G14 curve"""
    out = []
    for i in range(nb):
        g = torch.Generator("cpu").manual_seed(500 + i)
        out.append((torch.randn(b, 3, size, size, generator=g), torch.randn(b, 3, size, size, generator=g)))
    return out
