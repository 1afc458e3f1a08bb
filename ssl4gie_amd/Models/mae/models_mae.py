"""MaskedAutoencoderViT on the MI355X engine — drop-in for the reference class
(`/root/reference/Models/mae/models_mae.py:22-220`): same constructor signature, same attribute and
state_dict names (254 tensors for ViT-B), same `forward(imgs, mask_ratio) -> (loss, pred, mask)`.

What differs is where the arithmetic runs.  Nothing here calls a torch compute op on the hot
path: masking indices, patch gather, every GEMM, LayerNorm, attention, token (un)shuffles and the
loss gradient are HIP kernels behind libssl4gie_hip.so; both transformer stacks are executed by
the native block executor.  Two restructurings are arithmetic-preserving:
  * patches are embedded AFTER the keep-set is known, so the patch-embed GEMM only runs on the
    25 % of rows that survive (reference: embed all 196, then gather — models_mae.py:152-158);
  * decoder_pred's cls row is computed and ignored exactly as in the reference (:191-194).
Masking noise: drawn with torch.rand on the input's device like the reference (:129), or injected
with `noise=` (tests, bit-exact index parity); argsort ties are broken stably.
"""
from __future__ import annotations

from functools import partial

import torch
import torch.nn as nn

from ... import ops
from ...engine import DecoderAssembleFn, EngineModule, MaeLossFn, PatchEmbedFn
from ..vit_layers import Block, PatchEmbed
from .util.pos_embed import get_2d_sincos_pos_embed


class MaskedAutoencoderViT(EngineModule):
    """Masked Autoencoder with VisionTransformer backbone (engine-backed)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3,
                 embed_dim=1024, depth=24, num_heads=16,
                 decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16,
                 mlp_ratio=4., norm_layer=nn.LayerNorm, norm_pix_loss=False):
        super().__init__()
        self.in_chans = in_chans
        self.num_heads = num_heads
        self.decoder_num_heads = decoder_num_heads

        # encoder
        self.patch_embed = PatchEmbed(img_size, patch_size, in_chans, embed_dim)
        n = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, n + 1, embed_dim), requires_grad=False)
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio, qkv_bias=True,
                                           norm_layer=norm_layer) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)

        # decoder
        self.decoder_embed = nn.Linear(embed_dim, decoder_embed_dim, bias=True)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, decoder_embed_dim))
        self.decoder_pos_embed = nn.Parameter(torch.zeros(1, n + 1, decoder_embed_dim),
                                              requires_grad=False)
        self.decoder_blocks = nn.ModuleList([Block(decoder_embed_dim, decoder_num_heads, mlp_ratio,
                                                   qkv_bias=True, norm_layer=norm_layer)
                                             for _ in range(decoder_depth)])
        self.decoder_norm = norm_layer(decoder_embed_dim)
        self.decoder_pred = nn.Linear(decoder_embed_dim, patch_size ** 2 * in_chans, bias=True)

        self.norm_pix_loss = norm_pix_loss
        self.initialize_weights()

    # ------------------------------------------------------------------ init (reference :66-93)
    def initialize_weights(self):
        grid = int(self.patch_embed.num_patches ** .5)
        with torch.no_grad():
            for name in ("pos_embed", "decoder_pos_embed"):
                prm = getattr(self, name)
                table = get_2d_sincos_pos_embed(prm.shape[-1], grid, cls_token=True)
                prm.copy_(torch.from_numpy(table).float().unsqueeze(0))
        w = self.patch_embed.proj.weight.data
        nn.init.xavier_uniform_(w.view(w.shape[0], -1))  # conv initialised like a Linear
        nn.init.normal_(self.cls_token, std=.02)
        nn.init.normal_(self.mask_token, std=.02)
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.LayerNorm):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    # ------------------------------------------------------------------ small utilities
    def patchify(self, imgs):
        """[N, C, H, W] -> [N, L, p*p*C] ('nhwpqc'; reference :95-107) via the gather kernel."""
        p = self.patch_embed.patch_size[0]
        assert imgs.shape[2] == imgs.shape[3] and imgs.shape[2] % p == 0
        n, c = imgs.shape[0], imgs.shape[1]
        out = ops.patch_gather(imgs.contiguous().float(), p, order=1)
        return out.view(n, -1, p * p * c)

    def unpatchify(self, x):
        """Inverse of patchify (pure index permutation; reference :109-121)."""
        p = self.patch_embed.patch_size[0]
        h = w = int(x.shape[1] ** .5)
        assert h * w == x.shape[1]
        c = x.shape[2] // (p * p)
        return x.reshape(x.shape[0], h, w, p, p, c).permute(0, 5, 1, 3, 2, 4).reshape(
            x.shape[0], c, h * p, w * p)

    def _masking(self, n, mask_ratio, device, noise=None):
        L = self.patch_embed.num_patches
        keep = int(L * (1 - mask_ratio))
        if noise is None:
            noise = torch.rand(n, L, device=device)
        noise = noise.to(device=device, dtype=torch.float32).contiguous()
        assert noise.shape == (n, L)
        ids_shuffle, ids_restore, mask = ops.mask_argsort(noise, keep)
        return ids_shuffle, ids_restore, mask, keep

    def random_masking(self, x, mask_ratio, noise=None):
        """API-compatible with the reference (:123-148) for callers that mask token tensors
        themselves; the model's own forward masks before embedding instead."""
        n, L, d = x.shape
        ids_shuffle, ids_restore, mask, keep = self._masking(n, mask_ratio, x.device, noise)
        idx = ids_shuffle[:, :keep].unsqueeze(-1).expand(-1, -1, d)
        return torch.gather(x, 1, idx), mask, ids_restore

    # ------------------------------------------------------------------ forward pieces
    def forward_encoder(self, x, mask_ratio, noise=None):
        self._prepare()
        p = self.patch_embed.patch_size[0]
        assert x.shape[2] == self.patch_embed.img_size[0] and x.shape[3] == self.patch_embed.img_size[1]
        ids_shuffle, ids_restore, mask, keep = self._masking(x.shape[0], mask_ratio, x.device, noise)
        self._ids_shuffle = ids_shuffle
        tok = PatchEmbedFn.apply(x.float(), self.patch_embed.proj.weight, self.patch_embed.proj.bias,
                                 self.cls_token, self.pos_embed, ids_shuffle, keep, p, self.dtype_,
                                 self.sink(), self._lp)
        tok, _ = self._blocks(self.blocks, tok, self.num_heads, self.norm.eps)
        latent = self._ln(tok, self.norm)  # operand type; feeds decoder_embed directly
        return latent, mask, ids_restore

    def forward_decoder(self, x, ids_restore, _full=False):
        self._prepare()
        keep = x.shape[1] - 1
        ids_shuffle = getattr(self, "_ids_shuffle", None)
        if ids_shuffle is None or ids_shuffle.shape != ids_restore.shape:
            ids_shuffle = torch.argsort(ids_restore, dim=1)  # inverse permutation (plumbing)
        if x.dtype != self.dtype_:
            x = x.to(self.dtype_)
        y = self._linear(x, self.decoder_embed)
        xd = DecoderAssembleFn.apply(y, self.mask_token, self.decoder_pos_embed, ids_restore,
                                     ids_shuffle, keep, self.sink())
        xd, _ = self._blocks(self.decoder_blocks, xd, self.decoder_num_heads, self.decoder_norm.eps)
        h = self._ln(xd, self.decoder_norm)
        full = self._linear(h, self.decoder_pred, out_dtype=torch.float32)  # [N, 1+L, p*p*C]
        return full if _full else full[:, 1:, :]

    def forward_loss(self, imgs, pred, mask):
        """imgs [N, C, H, W]; pred [N, L, p*p*C] (or [N, 1+L, .] incl. the cls row);
        mask [N, L] with 1 = removed.  Mean squared error on removed patches (reference :198-214)."""
        p = self.patch_embed.patch_size[0]
        per_patch = MaeLossFn.apply(pred, imgs.contiguous().float(), mask, p,
                                    bool(self.norm_pix_loss))
        return per_patch.sum() / mask.sum()

    def forward(self, imgs, mask_ratio=0.75, noise=None):
        latent, mask, ids_restore = self.forward_encoder(imgs, mask_ratio, noise=noise)
        full = self.forward_decoder(latent, ids_restore, _full=True)
        loss = self.forward_loss(imgs, full, mask)
        return loss, full[:, 1:, :], mask


def _mae(embed_dim, depth, num_heads, patch_size=16, **kwargs):
    return MaskedAutoencoderViT(patch_size=patch_size, embed_dim=embed_dim, depth=depth,
                                num_heads=num_heads, decoder_embed_dim=512, decoder_depth=8,
                                decoder_num_heads=16, mlp_ratio=4,
                                norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def mae_vit_base_patch16_dec512d8b(**kwargs):
    return _mae(768, 12, 12, **kwargs)


def mae_vit_large_patch16_dec512d8b(**kwargs):
    return _mae(1024, 24, 16, **kwargs)


def mae_vit_huge_patch14_dec512d8b(**kwargs):
    return _mae(1280, 32, 16, patch_size=14, **kwargs)


# recommended archs (names used by main_pretrain.py:156 via models_mae.__dict__[args.model])
mae_vit_base_patch16 = mae_vit_base_patch16_dec512d8b
mae_vit_large_patch16 = mae_vit_large_patch16_dec512d8b
mae_vit_huge_patch14 = mae_vit_huge_patch14_dec512d8b
