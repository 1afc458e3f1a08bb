#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the REFERENCE's own classes.

Runs only in the authoring container (needs /root/reference, which never travels to the GPU
box).  The reference pins timm==0.6.12 without vendoring it, so its `models_mae.py` is imported
on top of `oracle.timm_restatement` registered under the `timm.*` names.  Weights
and inputs come from `oracle.synth` (pure functions of key/seed) so fixtures only hold inputs
that are not reproducible + expected outputs.

    python tests/golden/make_golden.py [--skip-curve]

Fixtures written:
  g1_masking.npz      random_masking (models_mae.py:123-148) on tie-free noise + crafted-tie row
  g2_patchify.npz     patchify / unpatchify (models_mae.py:95-121)
  g3_sincos.npz       fixed sin-cos tables (mae/util/pos_embed.py:20-67)
  g5_mae_tiny.npz     MaskedAutoencoderViT fwd + all grads, tiny config, norm_pix on/off
  g5_mae_vitb.npz     MaskedAutoencoderViT ViT-B fwd + grad norms / small grads, B=2
  g5_curve_tiny.npz   100-step AdamW loss curve, tiny config (B=8)
  g5_curve_vitb.npz   100-step AdamW loss curve, ViT-B (B=8)  [slow; --skip-curve]
  g_lr_sched.npz      per-iteration warm-up+cosine LR (mae/util/lr_sched.py:9-21)
  g6_dpt_depth.npz    reference DPT_decoder("depth") fwd + SSI loss + grads on seeded taps, B=2
  g7_ssi_loss.npz     reference ScaleAndShiftInvariantLoss(alpha=0.1) value + gradient
  g8_moco.npz         reference MoCo._build_mlp fwd/bwd, contrastive_loss (1-process gloo), LARS 3 steps
  g3_sincos.npz       (+ moco_768: VisionTransformerMoCo.build_2d_sincos_position_embedding, vits.py:53-69)
  g10_det.npz         reference Models/models.py detection backbone: VisionTransformer_from_Any(det=True)
                      trunk at 512^2 (WindowedAttention :155-210, _pos_embed_interp :310-323) fwd + grads,
                      ViT_from_MAE(det=True) trunk at 256^2, full backbone + ViTDet_FPN (:213-259) at 1024^2
  g11_vit_api.npz     reference ViT_from_MAE / ViT_from_MoCoV3 / VisionTransformer_from_Any: heads (cls,
                      spatial), dense taps (:427-456), ViT_from_MAE + DPT depth fwd + SSI loss + grad norms
  g12_resnet_dec.npz  reference ResNet_from_Any(dense="depth"): decode() (:16-60,128-135) on seeded stage
                      maps fwd + grads, and the whole model on top of oracle/torchvision_restatement.py
  g13_depth_curve.npz 60-step loss curve of the reference ViT_from_MAE(dense="depth") + SSI loss + AdamW(1e-4),
                      the statement sequence of train_depth.py:35-48 (BASELINE.json configs[3]), B = 2
  g14_moco_curve.npz  50-step loss curve of the reference MoCo_ResNet + LARS (builder.py:75-96, main_moco.py),
                      128 x 128 views, B = 16, 1-process gloo (BASELINE.json configs[2])
  g16_frozen.npz      frozen=True (linear probe / frozen finetune, models.py:138-142,341-345,459-463): ViT_from_MAE head,
                      ViT_from_MAE + DPT depth, ResNet_from_Any head — outputs, head / decoder gradients, no trunk gradients
  g18_eval_mode.npz   model.eval() inference (eval_*.py / predict_*.py): BatchNorm with running statistics after two
                      training-mode forwards — ResNet_from_Any dense + classifier, ViT_from_MAE(dense="seg")
  g17_bf16_bars.npz   per-tensor error of the reference's own bf16-autocast gradients against its fp64 gradients (G11
                      depth model, G10 det trunk): the bar the bf16 engine's whole-model gradients are held to
  g14_moco_fp64.npz   the same reference classes converted to double: 10-step loss curve, step-0 gradients and the
                      per-tensor error of the reference's own fp32 gradients against them (the parity gate of
                      tests/test_gpu_curves.py: the fp32 engine must be no further from fp64 than the reference's fp32)
"""
from __future__ import annotations

import argparse
import contextlib
import os
import sys
import types
from functools import partial

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import mae_ref, synth, timm_restatement, torchvision_restatement  # noqa: E402


def import_reference_mae():
    timm_restatement.install_as_timm()
    np.float = float  # pos_embed.py:56 uses the removed alias
    sys.path.insert(0, REF)
    import Models.mae.models_mae as ref_mae  # noqa
    return ref_mae


@contextlib.contextmanager
def injected_rand(noise: np.ndarray):
    """random_masking draws torch.rand(N, L) (models_mae.py:129); feed it our noise."""
    real = torch.rand
    t = torch.from_numpy(np.asarray(noise, dtype=np.float32))

    def fake(*shape, **kw):
        shp = tuple(shape[0]) if len(shape) == 1 and not isinstance(shape[0], int) else shape
        assert tuple(shp) == tuple(t.shape), (shp, t.shape)
        return t.clone()

    torch.rand = fake
    try:
        yield
    finally:
        torch.rand = real


def build_ref(ref_mae, cfg: mae_ref.MAEConfig, seed: int):
    import torch.nn as nn
    m = ref_mae.MaskedAutoencoderViT(
        img_size=cfg.img_size, patch_size=cfg.patch_size, in_chans=cfg.in_chans,
        embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads,
        decoder_embed_dim=cfg.decoder_embed_dim, decoder_depth=cfg.decoder_depth,
        decoder_num_heads=cfg.decoder_num_heads, mlp_ratio=cfg.mlp_ratio,
        norm_layer=partial(nn.LayerNorm, eps=cfg.ln_eps), norm_pix_loss=cfg.norm_pix_loss)
    sd = synth.mae_state_dict(cfg, seed)
    # the reference's own sincos tables must equal the oracle's before we overwrite them
    own = m.state_dict()
    assert set(own) == set(sd), set(own) ^ set(sd)
    for k in ("pos_embed", "decoder_pos_embed"):
        np.testing.assert_allclose(own[k].numpy(), sd[k].numpy(), rtol=0, atol=1e-6)
        sd[k] = own[k].clone()
    m.load_state_dict(sd, strict=True)
    return m, sd


def g1_masking(ref_mae):
    cfg = mae_ref.VIT_B
    m = ref_mae.MaskedAutoencoderViT.__new__(ref_mae.MaskedAutoencoderViT)  # method only
    noise = synth.synth_noise(8, 196, seed=0)
    x = torch.arange(8 * 196 * 4, dtype=torch.float32).reshape(8, 196, 4)
    with injected_rand(noise):
        xm, mask, ids_restore = ref_mae.MaskedAutoencoderViT.random_masking(m, x, 0.75)
    ids_shuffle = torch.argsort(torch.from_numpy(noise), dim=1)
    # crafted ties: expected output is the build's STABLE rule (oracle), not the reference
    tie = noise[:2].copy()
    tie[0, 5] = tie[0, 100] = tie[0, 17]
    tie[1, :] = 0.5
    o = mae_ref.masking_from_noise(tie, 0.75)
    np.savez_compressed(os.path.join(HERE, "g1_masking.npz"), noise=noise,
                        ids_shuffle=ids_shuffle.numpy(), ids_restore=ids_restore.numpy(),
                        mask=mask.numpy(), x_masked=xm.numpy(), x=x.numpy(),
                        tie_noise=tie, tie_ids_shuffle=o[0], tie_ids_restore=o[1],
                        tie_mask=o[3])
    print("g1 ok: kept/row", int((mask == 0).sum(1)[0]))


def g2_patchify(ref_mae):
    m = types.SimpleNamespace(patch_embed=types.SimpleNamespace(patch_size=(16, 16)))
    g = torch.Generator().manual_seed(7)
    imgs = torch.randn(2, 3, 64, 64, generator=g)
    p = ref_mae.MaskedAutoencoderViT.patchify(m, imgs)
    u = ref_mae.MaskedAutoencoderViT.unpatchify(m, p)
    assert torch.equal(u, imgs)
    big = synth.synth_images(1, mae_ref.VIT_B, seed=3)
    pb = ref_mae.MaskedAutoencoderViT.patchify(m, big)
    np.savez_compressed(os.path.join(HERE, "g2_patchify.npz"), imgs=imgs.numpy(),
                        patches=p.numpy(), big_rows=pb[0, [0, 13, 14, 195]].numpy())
    print("g2 ok")


def g3_sincos():
    sys.path.insert(0, REF)
    from Models.mae.util import pos_embed as pe
    np.float = float
    out = {f"mae_{d}": pe.get_2d_sincos_pos_embed(d, 14, cls_token=True).astype(np.float32)
           for d in (768, 512, 192, 128)}
    out["mae_192_g4"] = pe.get_2d_sincos_pos_embed(192, 4, cls_token=True).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "g3_sincos.npz"), **out)
    print("g3 ok")


def run_fwd_bwd(m, imgs, noise, mask_ratio=0.75):
    m.zero_grad(set_to_none=True)
    with injected_rand(noise):
        loss, pred, mask = m(imgs, mask_ratio=mask_ratio)
    loss.backward()
    grads = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
    return loss.detach(), pred.detach(), mask.detach(), grads


def g5_tiny(ref_mae):
    out = {}
    for npl in (False, True):
        cfg = mae_ref.MAEConfig(**{**mae_ref.TINY.__dict__, "norm_pix_loss": npl})
        m, sd = build_ref(ref_mae, cfg, seed=1)
        imgs = synth.synth_images(4, cfg, seed=1)
        noise = synth.synth_noise(4, cfg.num_patches, seed=1)
        loss, pred, mask, grads = run_fwd_bwd(m, imgs, noise)
        tag = "npl" if npl else "raw"
        out[f"{tag}/loss"] = loss.numpy()
        out[f"{tag}/pred"] = pred.numpy()
        out[f"{tag}/mask"] = mask.numpy()
        for k, g in grads.items():
            out[f"{tag}/grad/{k}"] = g.numpy()
        out[f"{tag}/digest"] = np.array(synth.state_dict_digest(sd))
        # oracle must agree with the reference before anything is written
        sdo = {k: v.clone().requires_grad_(v.is_floating_point() and "pos_embed" not in k)
               for k, v in sd.items()}
        lo, po, mo, _ = mae_ref.mae_forward(sdo, cfg, imgs, noise)
        np.testing.assert_allclose(lo.detach().numpy(), loss.numpy(), rtol=1e-5)
        np.testing.assert_allclose(po.detach().numpy(), pred.numpy(), rtol=1e-4, atol=1e-5)
        print(f"g5 tiny[{tag}] loss={float(loss):.6f} (oracle agrees)")
    np.savez_compressed(os.path.join(HERE, "g5_mae_tiny.npz"), **out)


def g5_vitb(ref_mae):
    cfg = mae_ref.MAEConfig(**{**mae_ref.VIT_B.__dict__, "norm_pix_loss": True})
    m, sd = build_ref(ref_mae, cfg, seed=0)
    imgs = synth.synth_images(2, cfg, seed=0)
    noise = synth.synth_noise(2, 196, seed=0)
    loss, pred, mask, grads = run_fwd_bwd(m, imgs, noise)
    out = {"loss": loss.numpy(), "pred": pred.numpy(), "mask": mask.numpy(),
           "digest": np.array(synth.state_dict_digest(sd)),
           "n_tensors": np.array(len(sd)),
           "n_params": np.array(sum(v.numel() for v in sd.values()))}
    names = sorted(grads)
    out["grad_names"] = np.array(names)
    out["grad_norms"] = np.array([float(grads[k].norm()) for k in names], dtype=np.float64)
    for k in names:
        g = grads[k]
        if g.numel() <= 4096:
            out[f"grad/{k}"] = g.numpy()
        else:
            out[f"gslice/{k}"] = g.reshape(g.shape[0], -1)[:8, :64].numpy()
    np.savez_compressed(os.path.join(HERE, "g5_mae_vitb.npz"), **out)
    print(f"g5 vitb loss={float(loss):.6f} tensors={len(sd)} params={int(out['n_params'])}")


def curve(ref_mae, cfg, b, steps, lr, fname):
    m, sd = build_ref(ref_mae, cfg, seed=0)
    from timm.optim import optim_factory
    groups = optim_factory.add_weight_decay(m, 0.05)  # main_pretrain.py:179
    opt = torch.optim.AdamW(groups, lr=lr, betas=(0.9, 0.95))  # main_pretrain.py:180
    losses = []
    for it in range(steps):
        imgs = synth.synth_images(b, cfg, seed=it % 4)
        noise = synth.synth_noise(b, cfg.num_patches, seed=100 + it)
        opt.zero_grad(set_to_none=True)
        with injected_rand(noise):
            loss, _, _ = m(imgs, mask_ratio=0.75)
        loss.backward()
        opt.step()
        losses.append(float(loss))
        if it % 10 == 0:
            print(f"  step {it}: {losses[-1]:.6f}", flush=True)
    np.savez_compressed(os.path.join(HERE, fname), losses=np.array(losses, dtype=np.float64),
                        lr=np.array(lr), batch=np.array(b), steps=np.array(steps),
                        weight_decay=np.array(0.05), betas=np.array([0.9, 0.95]),
                        digest=np.array(synth.state_dict_digest(sd)))
    print(fname, "first/last", losses[0], losses[-1])


def import_reference_models():
    """the reference's own Models/models.py (+ moco_v3/vits.py) on top of the timm / torchvision
    restatements (both pinned in requirements.txt:7,10, neither vendored nor installed)"""
    timm_restatement.install_as_timm()
    torchvision_restatement.install_as_torchvision()
    np.float = float
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import Models.models as ref_models  # noqa
    return ref_models


def load_keyed(m, seed, keep=()):
    """overwrite every tensor of the reference module (except `keep`: fixed tables it built itself)
    with oracle.synth.keyed_tensor(key, shape, seed); returns (shapes, digest)"""
    own = m.state_dict()
    sd = synth.keyed_state_dict({k: tuple(v.shape) for k, v in own.items()}, seed, keep=keep)
    for k, v in sd.items():
        assert v.dtype == own[k].dtype, (k, v.dtype, own[k].dtype)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert set(missing) <= set(keep) and not unexpected, (missing, unexpected)
    return {k: tuple(v.shape) for k, v in own.items()}, synth.state_dict_digest(m.state_dict())


def pack_grads(out, prefix, named_params, small=4096):
    grads = {k: p.grad for k, p in named_params if p.grad is not None}
    names = sorted(grads)
    out[prefix + "grad_names"] = np.array(names)
    out[prefix + "grad_norms"] = np.array([float(grads[k].double().norm()) for k in names], dtype=np.float64)
    for k in names:
        g = grads[k]
        if g.numel() <= small:
            out[f"{prefix}grad/{k}"] = g.numpy()
        else:
            out[f"{prefix}gslice/{k}"] = g.reshape(g.shape[0], -1)[:8, :64].numpy().copy()


def g3b_moco_sincos():
    import_reference_models()
    import Models.moco_v3.vits as ref_vits
    out = dict(np.load(os.path.join(HERE, "g3_sincos.npz")))
    for d, grid in ((768, 14), (384, 14), (192, 4)):
        obj = types.SimpleNamespace(patch_embed=types.SimpleNamespace(grid_size=(grid, grid)), embed_dim=d)
        ref_vits.VisionTransformerMoCo.build_2d_sincos_position_embedding(obj)
        t = obj.pos_embed.detach()
        assert t.shape == (1, grid * grid + 1, d) and not obj.pos_embed.requires_grad
        np.testing.assert_array_equal(mae_ref.sincos_2d_moco(d, grid).numpy(), t.numpy())
        out[f"moco_{d}" + ("" if grid == 14 else f"_g{grid}")] = t.numpy()
    np.savez_compressed(os.path.join(HERE, "g3_sincos.npz"), **out)
    print("g3b ok (oracle table bit-equal to the reference's)")


def g10_det():
    from oracle import det_ref
    rm = import_reference_models()
    out = {}
    # ---- (a) VisionTransformer_from_Any(det=True) trunk, 512^2: N = 1024 tokens, 4 windows of 256
    m = rm.VisionTransformer_from_Any(False, 0, False, None, True, 512, 768, 12, 12, "cls")
    assert "cls_token" not in m.state_dict()
    shapes, digest = load_keyed(m, seed=31)
    out["keys"] = np.array(sorted(shapes))
    out["digest"] = np.array(digest)
    g = torch.Generator("cpu").manual_seed(32)
    imgs = torch.randn(1, 3, 512, 512, generator=g)
    wgt = torch.randn(1, 1024, 768, generator=g)
    tok = m.forward_features(imgs)
    (tok * wgt).sum().backward()
    out["t512/tok_sub"] = tok.detach()[:, ::4].numpy()
    out["t512/tok_norm"] = np.array(float(tok.detach().double().norm()))
    pack_grads(out, "t512/", [(k, p) for k, p in m.named_parameters() if not k.startswith("fpn.")])
    # the oracle restatement must agree with the reference before anything is written
    sdo = {k: v.detach().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        ref_o = det_ref.det_trunk(sdo, imgs, 512)
    np.testing.assert_allclose(ref_o.numpy(), tok.detach().numpy(), rtol=1e-3, atol=2e-4)
    print(f"g10 trunk512: |tok|={float(out['t512/tok_norm']):.4f} (oracle agrees)")
    # ---- (b) the whole backbone + ViTDet_FPN at its hard-coded 1024^2 geometry, B = 1
    m.zero_grad(set_to_none=True)
    m.fixed_size = 1024
    m.patch_embed.img_size = (1024, 1024)
    g = torch.Generator("cpu").manual_seed(33)
    imgs = torch.randn(1, 3, 1024, 1024, generator=g)
    maps = m(imgs)
    assert list(maps.keys()) == ["0", "1", "2", "3", "pool"]
    loss = 0
    for i, (k, v) in enumerate(maps.items()):
        w = torch.randn(v.shape, generator=torch.Generator("cpu").manual_seed(40 + i))
        loss = loss + (v * w).sum() / v.numel() ** 0.5
        out[f"f1024/shape/{k}"] = np.array(v.shape)
        out[f"f1024/norm/{k}"] = np.array(float(v.detach().double().norm()))
        out[f"f1024/corner/{k}"] = v.detach()[:, :8, :16, :16].numpy().copy()
        out[f"f1024/center/{k}"] = v.detach()[:, 100:108, v.shape[2] // 2:v.shape[2] // 2 + 4,
                                              v.shape[3] // 2:v.shape[3] // 2 + 4].numpy().copy()
    loss.backward()
    out["f1024/loss"] = np.array(float(loss))
    pack_grads(out, "f1024/", list(m.named_parameters()))
    with torch.no_grad():
        tok_o = det_ref.det_trunk(sdo, imgs, 1024)
        maps_o = det_ref.fpn(sdo, tok_o)
    for k in maps:
        np.testing.assert_allclose(maps_o[k].numpy(), maps[k].detach().numpy(), rtol=2e-3, atol=2e-3)
    print(f"g10 full1024: loss={float(loss):.6f} (oracle agrees on all 5 maps)")
    del m
    # ---- (c) ViT_from_MAE(det=True) forward_encoder at 256^2 (one window; models.py:430-443)
    m = rm.ViT_from_MAE(None, False, 0, False, None, True, 256, 768, 12, 12, "cls")
    keep = ("pos_embed", "decoder_pos_embed")
    shapes, digest = load_keyed(m, seed=34, keep=keep)
    out["mae256/keys"] = np.array(sorted(shapes))
    out["mae256/digest"] = np.array(digest)
    imgs = torch.randn(2, 3, 256, 256, generator=torch.Generator("cpu").manual_seed(35))
    with torch.no_grad():
        tok = m.forward_encoder(imgs)
    out["mae256/tok"] = tok[:, ::2, ::2].numpy().copy()
    out["mae256/tok_norm"] = np.array(float(tok.double().norm()))
    np.savez_compressed(os.path.join(HERE, "g10_det.npz"), **out)
    print("g10 ok")


def g11_vit_api():
    from oracle import dpt_ref
    rm = import_reference_models()
    ref_loss = _load_by_path("ref_losses", os.path.join(REF, "Depth_estimation", "Metrics", "losses.py"))
    out = {}
    cfg = mae_ref.VIT_B
    imgs = synth.synth_images(2, cfg, seed=41)
    out["imgs_seed"] = np.array(41)
    keep = ("pos_embed", "decoder_pos_embed")
    # ---- ViT_from_MAE with a linear head: cls and spatial readouts (models.py:458-472)
    m = rm.ViT_from_MAE(None, True, 6, False, None, False, None, 768, 12, 12, "cls")
    assert "decoder_pos_embed" in m.state_dict() and "mask_token" not in m.state_dict()
    shapes, digest = load_keyed(m, seed=42, keep=keep)
    out["mae_head/keys"] = np.array(sorted(shapes)); out["mae_head/digest"] = np.array(digest)
    y = m(imgs)
    wy = torch.randn(y.shape, generator=torch.Generator("cpu").manual_seed(43))
    (y * wy).sum().backward()
    out["mae_head/cls"] = y.detach().numpy()
    pack_grads(out, "mae_head/", list(m.named_parameters()))
    m.out_token = "spatial"
    with torch.no_grad():
        out["mae_head/spatial"] = m(imgs).numpy()
        tok = m.forward_encoder(imgs)
    out["mae_head/tok_sub"] = tok[:, ::8].numpy().copy()
    sdo = {k: v.detach().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        np.testing.assert_allclose(mae_ref.vit_trunk(sdo, cfg, imgs, False).numpy(), tok.numpy(), rtol=1e-3, atol=1e-4)
    # ---- ViT_from_MAE(dense="depth"): taps after blocks 2/5/8/11 (no final norm), DPT decoder, SSI loss
    m = rm.ViT_from_MAE(None, False, 1, False, "depth", False, None, 768, 12, 12, "cls")
    shapes, digest = load_keyed(m, seed=44, keep=keep)
    out["mae_depth/keys"] = np.array(sorted(shapes)); out["mae_depth/digest"] = np.array(digest)
    taps = m.forward_encoder(imgs)
    assert isinstance(taps, list) and len(taps) == 4
    for i, t in enumerate(taps):
        out[f"mae_depth/tap_sub/{i}"] = t.detach()[:, ::16, ::4].numpy().copy()
        out[f"mae_depth/tap_norm/{i}"] = np.array(float(t.detach().double().norm()))
    g = torch.Generator("cpu").manual_seed(45)
    target = torch.rand(2, 1, 224, 224, generator=g)
    target = torch.where(torch.rand(2, 1, 224, 224, generator=g) < 0.1, torch.zeros(()), target)
    pred = m(imgs)
    loss = ref_loss.ScaleAndShiftInvariantLoss(alpha=0.1)(pred, target)
    loss.backward()
    out["mae_depth/pred_sub"] = pred.detach()[:, :, ::2, ::2].numpy().copy()
    out["mae_depth/pred_norm"] = np.array(float(pred.detach().double().norm()))
    out["mae_depth/loss"] = np.array(float(loss))
    pack_grads(out, "mae_depth/", list(m.named_parameters()))
    out["mae_depth/no_grad_params"] = np.array(sorted(k for k, p in m.named_parameters()
                                                      if p.requires_grad and p.grad is None))
    sdo = {k: v.detach().clone() for k, v in m.state_dict().items()}
    with torch.no_grad():
        taps_o = mae_ref.vit_trunk(sdo, cfg, imgs, True)
        pred_o = dpt_ref.dpt_forward({k[len("decoder."):]: v for k, v in sdo.items() if k.startswith("decoder.")},
                                     taps_o)
    np.testing.assert_allclose(pred_o.numpy(), pred.detach().numpy(), rtol=1e-3, atol=1e-5)
    print(f"g11 mae: depth loss={float(loss):.6f} (oracle trunk + DPT agree)")
    # ---- ViT_from_MoCoV3 (cat cls THEN add the fixed MoCo table, timm _pos_embed; models.py:545-578)
    m = rm.ViT_from_MoCoV3(None, True, 6, False, None, False, None, 768, "cls")
    assert not m.pos_embed.requires_grad and m.patch_embed.proj.weight.requires_grad
    shapes, digest = load_keyed(m, seed=46, keep=("pos_embed",))
    out["moco_head/keys"] = np.array(sorted(shapes)); out["moco_head/digest"] = np.array(digest)
    y = m(imgs)
    (y * wy).sum().backward()
    out["moco_head/cls"] = y.detach().numpy()
    pack_grads(out, "moco_head/", list(m.named_parameters()))
    m.out_token = "spatial"
    with torch.no_grad():
        out["moco_head/spatial"] = m(imgs).numpy()
    # ---- VisionTransformer_from_Any (learned pos_embed incl. the cls row; models.py:325-356)
    m = rm.VisionTransformer_from_Any(True, 12, False, None, False, None, 768, 12, 12, "cls")
    shapes, digest = load_keyed(m, seed=47)
    out["any_head/keys"] = np.array(sorted(shapes)); out["any_head/digest"] = np.array(digest)
    y = m(imgs)
    wy12 = torch.randn(y.shape, generator=torch.Generator("cpu").manual_seed(48))
    (y * wy12).sum().backward()
    out["any_head/cls"] = y.detach().numpy()
    pack_grads(out, "any_head/", list(m.named_parameters()))
    np.savez_compressed(os.path.join(HERE, "g11_vit_api.npz"), **out)
    print("g11 ok")


def resnet_stage_maps(seed, b=2, s=32):
    g = torch.Generator("cpu").manual_seed(seed)
    return [torch.relu(torch.randn(b, c, s >> i, s >> i, generator=g)) for i, c in enumerate((256, 512, 1024, 2048))]


def g12_resnet_dec():
    from oracle import dpt_ref, resnet_ref
    rm = import_reference_models()
    ref_loss = _load_by_path("ref_losses", os.path.join(REF, "Depth_estimation", "Metrics", "losses.py"))
    out = {}
    m = rm.ResNet_from_Any(None, False, 1, False, "depth")
    assert not any(k.startswith("fc.") for k in m.state_dict())
    shapes, digest = load_keyed(m, seed=51)
    out["keys"] = np.array(sorted(shapes)); out["digest"] = np.array(digest)
    m.train()
    # ---- (a) decode() on seeded stage maps: the reference's own decoder code (models.py:16-60,128-135)
    maps = [t.requires_grad_(True) for t in resnet_stage_maps(52)]
    g = torch.Generator("cpu").manual_seed(53)
    target = torch.rand(2, 1, 128, 128, generator=g)
    target = torch.where(torch.rand(2, 1, 128, 128, generator=g) < 0.1, torch.zeros(()), target)
    pred = m.decode(maps)
    loss = ref_loss.ScaleAndShiftInvariantLoss(alpha=0.1)(pred, target)
    loss.backward()
    out["dec/pred"] = pred.detach().numpy()
    out["dec/loss"] = np.array(float(loss))
    pack_grads(out, "dec/", [(k, p) for k, p in m.named_parameters() if p.grad is not None])
    for i, t in enumerate(maps):
        out[f"dec/map_grad_norm/{i}"] = np.array(float(t.grad.double().norm()))
        out[f"dec/map_grad_slice/{i}"] = t.grad[:, :16, :4, :4].numpy().copy()
    out["dec/running_mean/decoder_levels.0.chan_reduce.1"] = m.decoder_levels[0].chan_reduce[1].running_mean.numpy().copy()
    out["dec/running_var/decoder_levels.2.blocks.2.process.7"] = m.decoder_levels[2].blocks[2].process[7].running_var.numpy().copy()
    # oracle (fp64) must agree with the reference's decoder before anything is written
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in synth.keyed_state_dict(shapes, 51).items()}
    with torch.no_grad():
        mo = [t.detach().double() for t in maps]
        o = resnet_ref.dec_level(sd64, "decoder_levels.0", mo[-1], mo[-2])
        o = resnet_ref.dec_level(sd64, "decoder_levels.1", o, mo[-3])
        o = resnet_ref.dec_level(sd64, "decoder_levels.2", o, mo[-4])
        pred_o = resnet_ref.output_head(sd64, o)
    np.testing.assert_allclose(pred_o.float().numpy(), pred.detach().numpy(), rtol=1e-3, atol=1e-5)
    print(f"g12 decode: loss={float(loss):.6f} (oracle decoder agrees)")
    # ---- (b) the whole model (trunk = torchvision restatement, unpinned at that boundary)
    m.zero_grad(set_to_none=True)
    load_keyed(m, seed=51)  # reset the running statistics
    imgs = torch.randn(4, 3, 128, 128, generator=torch.Generator("cpu").manual_seed(54))
    target = torch.rand(4, 1, 128, 128, generator=torch.Generator("cpu").manual_seed(55))
    pred = m(imgs)
    loss = ref_loss.ScaleAndShiftInvariantLoss(alpha=0.1)(pred, target)
    loss.backward()
    out["full/pred"] = pred.detach().numpy()
    out["full/loss"] = np.array(float(loss))
    pack_grads(out, "full/", list(m.named_parameters()), small=2048)
    with torch.no_grad():
        pred_o = resnet_ref.resnet50_dense(sd64, imgs.double())
    np.testing.assert_allclose(pred_o.float().numpy(), pred.detach().numpy(), rtol=2e-3, atol=1e-4)
    # ---- (c) classification path: pooled features + linear head (models.py:143-149)
    m2 = rm.ResNet_from_Any(None, True, 6, False, None)
    shapes2, digest2 = load_keyed(m2, seed=56)
    out["cls/keys"] = np.array(sorted(shapes2)); out["cls/digest"] = np.array(digest2)
    m2.train()
    y = m2(imgs)
    out["cls/logits"] = y.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "g12_resnet_dec.npz"), **out)
    print(f"g12 ok: full loss={float(out['full/loss']):.6f}")


def g_lr_sched():
    sys.path.insert(0, REF)
    from Models.mae.util import lr_sched
    args = types.SimpleNamespace(lr=1.5e-4 * 4096 / 256, min_lr=0.0, warmup_epochs=40, epochs=800)
    class Opt:  # noqa
        param_groups = [{"lr": 0.0}, {"lr": 0.0, "lr_scale": 0.5}]
    epochs = np.concatenate([np.linspace(0, 45, 91), np.linspace(100, 800, 15)])
    lrs, lrs_scaled = [], []
    for e in epochs:
        lr_sched.adjust_learning_rate(Opt, float(e), args)
        lrs.append(Opt.param_groups[0]["lr"])
        lrs_scaled.append(Opt.param_groups[1]["lr"])
    np.savez_compressed(os.path.join(HERE, "g_lr_sched.npz"), epochs=epochs, lr=np.array(lrs),
                        lr_scaled=np.array(lrs_scaled), base_lr=np.array(args.lr),
                        min_lr=np.array(0.0), warmup_epochs=np.array(40),
                        total_epochs=np.array(800))
    print("lr_sched ok")


def _load_by_path(name, path):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def dpt_inputs(seed, b=2):
    g = torch.Generator("cpu").manual_seed(seed)
    acts = [torch.randn(b, 197, 768, generator=g) for _ in range(4)]
    target = torch.rand(b, 1, 224, 224, generator=g)
    target = torch.where(torch.rand(b, 1, 224, 224, generator=g) < 0.1, torch.zeros(()), target)
    return acts, target


def g6_dpt():
    """the reference's DPT_decoder (pure torch, imported as is) + its SSI loss, fwd and bwd"""
    from oracle import dpt_ref
    ref_dpt = _load_by_path("ref_dpt", os.path.join(REF, "Models", "DPT_decoder.py"))
    ref_loss = _load_by_path("ref_losses", os.path.join(REF, "Depth_estimation", "Metrics", "losses.py"))
    sd = dpt_ref.dpt_state_dict(seed=3)
    m = ref_dpt.DPT_decoder(num_classes=1, dense="depth")
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == dpt_ref.dpt_param_shapes()
    m.load_state_dict(sd, strict=True)
    acts, target = dpt_inputs(11)
    acts = [a.requires_grad_(True) for a in acts]
    skips = m.forward_skip([a for a in acts])
    out = m(acts)
    loss = ref_loss.ScaleAndShiftInvariantLoss(alpha=0.1)(out, target)
    loss.backward()
    # the oracle must agree with the reference before anything is written
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    acts_o = [a.detach().clone().requires_grad_(True) for a in acts]
    out_o, mid = dpt_ref.dpt_forward(sdo, acts_o, return_all=True)
    loss_o = dpt_ref.ssi_loss(out_o, target, alpha=0.1)
    loss_o.backward()
    np.testing.assert_allclose(out_o.detach().numpy(), out.detach().numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(float(loss_o), float(loss), rtol=1e-5)
    np.testing.assert_allclose(mid["layer_4"].detach().numpy(), skips[3].detach().numpy(), rtol=1e-4, atol=1e-5)
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    missing = sorted(k for k, p in m.named_parameters() if p.grad is None)
    out_d = {"out": out.detach().numpy(), "loss": np.array(float(loss)),
             "layer_4": skips[3].detach().numpy(), "layer_1_slice": skips[0].detach().numpy()[:, :8, :8, :8],
             "no_grad_params": np.array(missing), "seed_weights": np.array(3), "seed_inputs": np.array(11),
             "grad_names": np.array(sorted(grads)),
             "grad_norms": np.array([float(grads[k].norm()) for k in sorted(grads)], dtype=np.float64)}
    for k, g in grads.items():
        if g.numel() <= 4096:
            out_d[f"grad/{k}"] = g.numpy()
        else:
            out_d[f"gslice/{k}"] = g.reshape(g.shape[0], -1)[:8, :64].numpy()
    for i, a in enumerate(acts):
        out_d[f"act_grad_norm/{i}"] = np.array(float(a.grad.norm()))
        out_d[f"act_grad_slice/{i}"] = a.grad[:, :4, :64].numpy()
    np.savez_compressed(os.path.join(HERE, "g6_dpt_depth.npz"), **out_d)
    print(f"g6 dpt: loss={float(loss):.6f} out mean={float(out.mean()):.4f}; params without grad: {missing}")


def g9_dpt_seg():
    """the reference's DPT_decoder(dense="seg") (BatchNorm fusion blocks) + its SoftDiceLoss, fwd and
    bwd in training mode; Dropout.p is set to 0 (its mask is RNG state, not arithmetic)"""
    from oracle import dpt_ref
    ref_dpt = _load_by_path("ref_dpt", os.path.join(REF, "Models", "DPT_decoder.py"))
    ref_loss = _load_by_path("ref_seg_losses", os.path.join(REF, "Binary_segmentation", "Metrics", "losses.py"))
    sd = dpt_ref.seg_state_dict(seed=5)
    m = ref_dpt.DPT_decoder(num_classes=1, dense="seg")
    ref_shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()
                  if "running" not in k and "num_batches" not in k}
    assert ref_shapes == dpt_ref.seg_param_shapes(), set(ref_shapes) ^ set(dpt_ref.seg_param_shapes())
    m.load_state_dict(sd, strict=False)
    m.output_conv[3].p = 0.0
    m.train()
    acts, _ = dpt_inputs(13)
    g = torch.Generator("cpu").manual_seed(14)
    target = (torch.rand(2, 1, 224, 224, generator=g) < 0.3).float()
    acts = [a.requires_grad_(True) for a in acts]
    out = m(acts)
    loss = ref_loss.SoftDiceLoss()(out, target)
    loss.backward()
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    acts_o = [a.detach().clone().requires_grad_(True) for a in acts]
    out_o = dpt_ref.seg_forward(sdo, acts_o)
    loss_o = dpt_ref.soft_dice_loss(out_o, target)
    loss_o.backward()
    np.testing.assert_allclose(out_o.detach().numpy(), out.detach().numpy(), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(float(loss_o), float(loss), rtol=1e-5)
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    missing = sorted(k for k, p in m.named_parameters() if p.grad is None)
    for k in ("layer1_rn.weight", "refinenet1.resConfUnit2.bn2.weight", "output_conv.4.weight"):
        err = float((sdo[k].grad - grads[k]).norm() / grads[k].norm())  # fp32 order-of-summation noise
        assert err < 5e-3, (k, err)
    out_d = {"out": out.detach().numpy(), "loss": np.array(float(loss)), "target": target.numpy().astype(np.uint8),
             "no_grad_params": np.array(missing), "seed_weights": np.array(5), "seed_inputs": np.array(13),
             "state_dict_keys": np.array(sorted(m.state_dict().keys())),
             "grad_names": np.array(sorted(grads)),
             "grad_norms": np.array([float(grads[k].norm()) for k in sorted(grads)], dtype=np.float64),
             "running_mean/output_conv.1": m.output_conv[1].running_mean.numpy()}
    for k, gr in grads.items():
        if gr.numel() <= 4096:
            out_d[f"grad/{k}"] = gr.numpy()
    for i, a in enumerate(acts):
        out_d[f"act_grad_norm/{i}"] = np.array(float(a.grad.norm()))
        out_d[f"act_grad_slice/{i}"] = a.grad[:, :4, :64].numpy()
    np.savez_compressed(os.path.join(HERE, "g9_dpt_seg.npz"), **out_d)
    print(f"g9 dpt seg: loss={float(loss):.6f} out mean={float(out.mean()):.4f}; params without grad: {missing}")


def g7_ssi():
    ref_loss = _load_by_path("ref_losses", os.path.join(REF, "Depth_estimation", "Metrics", "losses.py"))
    g = torch.Generator("cpu").manual_seed(5)
    pred = torch.rand(3, 1, 64, 64, generator=g).requires_grad_(True)
    target = torch.rand(3, 1, 64, 64, generator=g)
    target = torch.where(torch.rand(3, 1, 64, 64, generator=g) < 0.2, torch.zeros(()), target)
    target[2] = 0  # an image without valid pixels (empty mask, singular system)
    loss = ref_loss.ScaleAndShiftInvariantLoss(alpha=0.1)(pred, target)
    loss.backward()
    np.savez_compressed(os.path.join(HERE, "g7_ssi_loss.npz"), pred=pred.detach().numpy(),
                        target=target.numpy(), loss=np.array(float(loss)), grad=pred.grad.numpy())
    print(f"g7 ssi: loss={float(loss):.6f}")


def g8_moco():
    """the reference's own MoCo glue: MLP builder, InfoNCE loss (in a 1-process gloo group; the only
    shim is Tensor.cuda -> identity because this container has no GPU) and LARS"""
    import torch.distributed as dist
    ref_b = _load_by_path("ref_builder", os.path.join(REF, "Models", "moco_v3", "moco", "builder.py"))
    ref_o = _load_by_path("ref_lars", os.path.join(REF, "Models", "moco_v3", "moco", "optimizer.py"))
    out = {}
    g = torch.Generator("cpu").manual_seed(21)
    obj = ref_b.MoCo.__new__(ref_b.MoCo)  # _build_mlp does not touch self
    for tag, args in (("proj2", (2, 64, 128, 32)), ("pred2", (2, 32, 128, 32, False)), ("proj3", (3, 48, 96, 32))):
        mlp = ref_b.MoCo._build_mlp(obj, *args)
        with torch.no_grad():
            for prm in mlp.parameters():
                prm.copy_(torch.randn(prm.shape, generator=g) * (0.2 if prm.dim() > 1 else 1.0))
        x = torch.randn(16, args[1], generator=g).requires_grad_(True)
        y = mlp(x)
        dy = torch.randn(y.shape, generator=g)
        y.backward(dy)
        for k, v in mlp.state_dict().items():
            if "running" not in k and "num_batches" not in k:
                out[f"{tag}/sd/{k}"] = v.numpy()
        out[f"{tag}/keys"] = np.array(list(mlp.state_dict().keys()))
        out[f"{tag}/x"] = x.detach().numpy(); out[f"{tag}/y"] = y.detach().numpy()
        out[f"{tag}/dy"] = dy.numpy(); out[f"{tag}/dx"] = x.grad.numpy()
        for k, prm in mlp.named_parameters():
            out[f"{tag}/grad/{k}"] = prm.grad.numpy()
    # contrastive loss from the reference method itself
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=0, world_size=1)
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        obj.T = 0.2
        q = torch.randn(12, 32, generator=g).requires_grad_(True)
        k = torch.randn(12, 32, generator=g)
        loss = ref_b.MoCo.contrastive_loss(obj, q, k)
        loss.backward()
        out["cl/q"] = q.detach().numpy(); out["cl/k"] = k.numpy(); out["cl/T"] = np.array(0.2)
        out["cl/loss"] = np.array(float(loss)); out["cl/dq"] = q.grad.numpy()
    finally:
        torch.Tensor.cuda = real_cuda
        dist.destroy_process_group()
    # LARS: 3 steps on a matrix, a bias vector and a zero matrix
    ps = [torch.nn.Parameter(torch.randn(8, 16, generator=g)), torch.nn.Parameter(torch.randn(16, generator=g)),
          torch.nn.Parameter(torch.zeros(4, 4))]
    opt = ref_o.LARS(ps, lr=0.3, weight_decay=1e-2, momentum=0.9)
    for i, prm in enumerate(ps):
        out[f"lars/p0/{i}"] = prm.detach().numpy().copy()
    for step in range(3):
        for i, prm in enumerate(ps):
            prm.grad = torch.randn(prm.shape, generator=g)
            out[f"lars/g{step}/{i}"] = prm.grad.numpy().copy()
        opt.step()
        for i, prm in enumerate(ps):
            out[f"lars/p{step + 1}/{i}"] = prm.detach().numpy().copy()
    np.savez_compressed(os.path.join(HERE, "g8_moco.npz"), **out)
    print(f"g8 moco: contrastive loss={float(out['cl/loss']):.6f}")


def g13_depth_curve(steps=60):
    """BASELINE.json configs[3]: the reference's own ViT_from_MAE(dense="depth") (models.py:458-475) trained with
    its own ScaleAndShiftInvariantLoss(alpha=0.1) and AdamW(lr 2e-5) exactly as train_depth.py:35-48,230,280 does
    (zero_grad / forward / loss / backward / step; fp32 on the CPU, no autocast), B = 2, four seeded batches in
    rotation.  Weights: the keyed set of G11's depth model (seed 44)."""
    rm = import_reference_models()
    ref_loss = _load_by_path("ref_losses", os.path.join(REF, "Depth_estimation", "Metrics", "losses.py"))
    m = rm.ViT_from_MAE(None, False, 1, False, "depth", False, None, 768, 12, 12, "cls")
    shapes, digest = load_keyed(m, seed=44, keep=("pos_embed", "decoder_pos_embed"))
    m.train()
    opt = torch.optim.AdamW(m.parameters(), lr=2e-5)      # train_depth.py:230 (--learning-rate; at the default 1e-4
    # this random-weight model saturates its Sigmoid head within three steps and the curve degenerates)
    loss_fn = ref_loss.ScaleAndShiftInvariantLoss(alpha=0.1)  # train_depth.py:280
    batches = synth.depth_batches()
    losses = []
    for it in range(steps):
        data, target = batches[it % len(batches)]
        opt.zero_grad()
        loss = loss_fn(m(data), target)
        loss.backward()
        opt.step()
        losses.append(float(loss))
        if it % 10 == 0:
            print(f"  g13 step {it}: {losses[-1]:.6f}", flush=True)
    np.savez_compressed(os.path.join(HERE, "g13_depth_curve.npz"), losses=np.array(losses, dtype=np.float64),
                        keys=np.array(sorted(shapes)), digest=np.array(digest), lr=np.array(2e-5),
                        batch=np.array(2), steps=np.array(steps), n_batches=np.array(len(batches)))
    print("g13 first/last", losses[0], losses[-1])


def g14_moco_curve(steps=50):
    """BASELINE.json configs[2]: the reference's own MoCo_ResNet (moco/builder.py:11-110; torchvision's resnet50
    through oracle/torchvision_restatement.py, zero_init_residual as main_moco.py builds it) trained with its own
    LARS (moco/optimizer.py), momentum 0.99, T = 1.0, 128 x 128 views, B = 16 (at 64 x 64 / B = 8 the random-init
    network is chaotic: 1e-4 relative weight differences move the loss by 2 %), in a 1-process gloo group (the only
    shim: Tensor.cuda -> identity, no GPU here).  Weights: keyed set (seed 61) with the momentum encoder copied
    from the base encoder as MoCo.__init__ does."""
    import torch.distributed as dist
    from functools import partial
    import_reference_models()  # installs the torchvision restatement
    import torchvision.models as tvm
    ref_b = _load_by_path("ref_builder", os.path.join(REF, "Models", "moco_v3", "moco", "builder.py"))
    ref_o = _load_by_path("ref_lars", os.path.join(REF, "Models", "moco_v3", "moco", "optimizer.py"))
    torch.manual_seed(0)
    m = ref_b.MoCo_ResNet(partial(tvm.resnet50, zero_init_residual=True), 256, 1024, 1.0)
    shapes, digest = load_keyed(m, seed=61)
    with torch.no_grad():
        for pb, pm in zip(m.base_encoder.parameters(), m.momentum_encoder.parameters()):
            pm.copy_(pb)
    digest = synth.state_dict_digest(m.state_dict())
    m.train()
    opt = ref_o.LARS(m.parameters(), lr=0.02, weight_decay=1e-6, momentum=0.9)  # main_moco.py:213-216
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
    dist.init_process_group("gloo", rank=0, world_size=1)
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    views = synth.moco_views(b=16, size=128)
    losses, extra = [], {}
    try:
        for it in range(steps):
            x1, x2 = views[it % len(views)]
            loss = m(x1, x2, 0.99)          # main_moco.py:336
            opt.zero_grad()
            loss.backward()
            if it == 0:  # every gradient of the first step (norms; small ones in full, slices of the large ones)
                pack_grads(extra, "step0/", [(k, p) for k, p in m.named_parameters() if p.requires_grad])
            opt.step()
            losses.append(float(loss))
            if it % 10 == 0:
                print(f"  g14 step {it}: {losses[-1]:.6f}", flush=True)
    finally:
        torch.Tensor.cuda = real_cuda
        dist.destroy_process_group()
    np.savez_compressed(os.path.join(HERE, "g14_moco_curve.npz"), losses=np.array(losses, dtype=np.float64),
                        keys=np.array(sorted(shapes)), digest=np.array(digest), lr=np.array(0.02),
                        batch=np.array(16), steps=np.array(steps), size=np.array(128), n_batches=np.array(len(views)),
                        **extra)
    print("g14 first/last", losses[0], losses[-1])


def g14_moco_fp64(steps=10):
    """The fp64 evaluation of G14's first steps: the SAME reference classes (MoCo_ResNet, LARS) as g14_moco_curve,
    converted to double (`m.double()`, double views; LARS on double parameters), same keyed weights, same views —
    the ground truth both fp32 arithmetics (the reference's own CPU fp32 of G14 and the engine's) are measured
    against by tests/test_gpu_curves.py: "no worse than the reference's own fp32".  Stores the 10-step loss curve,
    every gradient of step 0 (norms, small tensors in full, slices of the large ones, as G14) and, per tensor, the
    reference-fp32 error against it (relative L2, the measure of tools/g14_conditioning.py)."""
    import torch.distributed as dist
    from functools import partial
    import_reference_models()
    import torchvision.models as tvm
    ref_b = _load_by_path("ref_builder", os.path.join(REF, "Models", "moco_v3", "moco", "builder.py"))
    ref_o = _load_by_path("ref_lars", os.path.join(REF, "Models", "moco_v3", "moco", "optimizer.py"))
    g14 = np.load(os.path.join(HERE, "g14_moco_curve.npz"))
    torch.manual_seed(0)
    m = ref_b.MoCo_ResNet(partial(tvm.resnet50, zero_init_residual=True), 256, 1024, 1.0)
    load_keyed(m, seed=61)
    with torch.no_grad():
        for pb, pm in zip(m.base_encoder.parameters(), m.momentum_encoder.parameters()):
            pm.copy_(pb)
    assert synth.state_dict_digest(m.state_dict()) == str(g14["digest"])  # the weights G14 ran with, then widened
    m.double()
    m.train()
    opt = ref_o.LARS(m.parameters(), lr=0.02, weight_decay=1e-6, momentum=0.9)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29535")
    dist.init_process_group("gloo", rank=0, world_size=1)
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    views = synth.moco_views(b=16, size=128)
    losses, extra = [], {}
    try:
        for it in range(steps):
            x1, x2 = views[it % len(views)]
            loss = m(x1.double(), x2.double(), 0.99)
            opt.zero_grad()
            loss.backward()
            if it == 0:
                pack_grads(extra, "step0/", [(k, p) for k, p in m.named_parameters() if p.requires_grad])
            opt.step()
            losses.append(float(loss))
            print(f"  g14-fp64 step {it}: {losses[-1]:.9f}   (reference fp32: {float(g14['losses'][it]):.9f})", flush=True)
    finally:
        torch.Tensor.cuda = real_cuda
        dist.destroy_process_group()
    # per tensor: the reference's own fp32 gradient (G14) against this one
    names = extra["step0/grad_names"].tolist()
    assert names == g14["step0/grad_names"].tolist()
    ref_err = []
    for k in names:
        key = f"step0/grad/{k}" if f"step0/grad/{k}" in extra else f"step0/gslice/{k}"
        a, b = np.asarray(g14[key], dtype=np.float64), np.asarray(extra[key], dtype=np.float64)
        ref_err.append(float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300)))
    np.savez_compressed(os.path.join(HERE, "g14_moco_fp64.npz"), losses=np.array(losses, dtype=np.float64),
                        ref_fp32_losses=np.asarray(g14["losses"][:steps], dtype=np.float64),
                        ref_fp32_grad_err=np.array(ref_err, dtype=np.float64), digest=np.array(str(g14["digest"])),
                        steps=np.array(steps), **extra)
    e = np.array(ref_err)
    print(f"g14-fp64: reference fp32 vs fp64 gradients: median {np.median(e):.3e} max {e.max():.3e}; "
          f"losses fp64 {losses[0]:.9f} .. {losses[-1]:.9f}")


def g16_frozen():
    """frozen=True of the reference's three wrappers (models.py:138-142, 341-345, 459-463; CLI --frozen): the trunk
    runs under no_grad — in training mode, so a ResNet trunk still uses and updates batch statistics — and only the
    head / decoder receive gradients."""
    rm = import_reference_models()
    ref_loss = _load_by_path("ref_losses", os.path.join(REF, "Depth_estimation", "Metrics", "losses.py"))
    out = {}
    cfg = mae_ref.VIT_B
    imgs = synth.synth_images(2, cfg, seed=71)
    out["imgs_seed"] = np.array(71)
    keep = ("pos_embed", "decoder_pos_embed")
    # ---- ViT_from_MAE(head=True, frozen=True): linear probe
    m = rm.ViT_from_MAE(None, True, 6, True, None, False, None, 768, 12, 12, "cls")
    shapes, digest = load_keyed(m, seed=72, keep=keep)
    out["mae_head/keys"] = np.array(sorted(shapes)); out["mae_head/digest"] = np.array(digest)
    m.train()
    y = m(imgs)
    wy = torch.randn(y.shape, generator=torch.Generator("cpu").manual_seed(73))
    (y * wy).sum().backward()
    out["mae_head/cls"] = y.detach().numpy()
    got = sorted(k for k, p in m.named_parameters() if p.grad is not None)
    assert got == ["lin_head.bias", "lin_head.weight"], got
    pack_grads(out, "mae_head/", list(m.named_parameters()), small=8192)
    # ---- ViT_from_MAE(dense="depth", frozen=True): frozen trunk, DPT decoder trained
    m = rm.ViT_from_MAE(None, False, 1, True, "depth", False, None, 768, 12, 12, "cls")
    shapes, digest = load_keyed(m, seed=74, keep=keep)
    out["mae_depth/keys"] = np.array(sorted(shapes)); out["mae_depth/digest"] = np.array(digest)
    m.train()
    g = torch.Generator("cpu").manual_seed(75)
    target = torch.rand(2, 1, 224, 224, generator=g)
    target = torch.where(torch.rand(2, 1, 224, 224, generator=g) < 0.1, torch.zeros(()), target)
    pred = m(imgs)
    loss = ref_loss.ScaleAndShiftInvariantLoss(alpha=0.1)(pred, target)
    loss.backward()
    out["mae_depth/pred_sub"] = pred.detach()[:, :, ::2, ::2].numpy().copy()
    out["mae_depth/loss"] = np.array(float(loss))
    with_grad = sorted(k for k, p in m.named_parameters() if p.grad is not None)
    assert with_grad and all(k.startswith("decoder.") for k in with_grad), with_grad[:4]
    out["mae_depth/with_grad"] = np.array(with_grad)
    pack_grads(out, "mae_depth/", list(m.named_parameters()))
    # ---- ResNet_from_Any(head=True, frozen=True): training-mode BatchNorm inside the no_grad trunk
    m = rm.ResNet_from_Any(None, True, 6, True, None)
    shapes, digest = load_keyed(m, seed=76)
    out["resnet_head/keys"] = np.array(sorted(shapes)); out["resnet_head/digest"] = np.array(digest)
    m.train()
    ximgs = torch.randn(4, 3, 128, 128, generator=torch.Generator("cpu").manual_seed(77))
    y = m(ximgs)
    wy = torch.randn(y.shape, generator=torch.Generator("cpu").manual_seed(78))
    (y * wy).sum().backward()
    out["resnet_head/logits"] = y.detach().numpy()
    got = sorted(k for k, p in m.named_parameters() if p.grad is not None)
    assert got == ["lin_head.bias", "lin_head.weight"], got
    pack_grads(out, "resnet_head/", list(m.named_parameters()), small=16384)
    out["resnet_head/running_mean/bn1"] = m.bn1.running_mean.numpy().copy()
    out["resnet_head/running_var/layer4.2.bn3"] = m.layer4[2].bn3.running_var.numpy().copy()
    np.savez_compressed(os.path.join(HERE, "g16_frozen.npz"), **out)
    print(f"g16 ok: depth loss {float(out['mae_depth/loss']):.6f}")


def _sample(t, n=2048):
    """a strided sample of a tensor's elements (the whole tensor when it is small)"""
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n]


def g17_bf16_bars():
    """Per-tensor error bars for the bf16 engine's whole-model gradients: the SAME reference
    classes and keyed weights as G11 depth / G10 t512, evaluated (a) in double — the ground truth — and (b) under
    the reference's own mixed-precision policy on this host, torch.autocast("cpu", bfloat16) (train_depth.py:41-46
    uses torch.cuda.amp.autocast: half-precision matmul / conv, fp32 LayerNorm / softmax / loss).  Stored per tensor:
    a strided sample of the double gradient, its norm, and the relative L2 error of the autocast gradient on that
    sample.  tests/test_gpu_models_golden.py gates the bf16 engine at 1.5 x that distribution (median and worst)."""
    rm = import_reference_models()
    ref_loss = _load_by_path("ref_losses", os.path.join(REF, "Depth_estimation", "Metrics", "losses.py"))
    out = {}

    def run(tag, build, seed, keep, step):
        res = {}
        for mode in ("fp64", "bf16"):
            m = build()
            load_keyed(m, seed=seed, keep=keep)
            m.train()
            if mode == "fp64":
                m.double()
            loss = step(m, mode)
            loss.backward()
            res[mode] = ({k: p.grad.detach().double() for k, p in m.named_parameters() if p.grad is not None},
                         float(loss))
            del m
        g64, l64 = res["fp64"]
        g16, l16 = res["bf16"]
        names = sorted(g64)
        assert names == sorted(g16)
        out[f"{tag}/names"] = np.array(names)
        out[f"{tag}/loss_fp64"] = np.array(l64)
        out[f"{tag}/loss_autocast"] = np.array(l16)
        errs = []
        for k in names:
            a, b = _sample(g64[k]), _sample(g16[k])
            out[f"{tag}/sample/{k}"] = a.float().numpy().copy()
            errs.append(float((a - b).norm() / (a.norm() + 1e-300)))
        out[f"{tag}/autocast_err"] = np.array(errs, dtype=np.float64)
        out[f"{tag}/norm_fp64"] = np.array([float(g64[k].norm()) for k in names], dtype=np.float64)
        e = np.array(errs)
        print(f"g17 {tag}: loss fp64 {l64:.6f} autocast {l16:.6f}; autocast gradient error median {np.median(e):.3e} "
              f"p90 {np.quantile(e, 0.9):.3e} max {e.max():.3e} ({names[int(e.argmax())]})", flush=True)

    # ---- G11's ViT_from_MAE(dense="depth") + SSI loss (seed 44, images 41, target 45)
    cfg = mae_ref.VIT_B
    imgs = synth.synth_images(2, cfg, seed=41)
    g = torch.Generator("cpu").manual_seed(45)
    target = torch.rand(2, 1, 224, 224, generator=g)
    target = torch.where(torch.rand(2, 1, 224, 224, generator=g) < 0.1, torch.zeros(()), target)

    def depth_step(m, mode):
        if mode == "fp64":
            return ref_loss.ScaleAndShiftInvariantLoss(alpha=0.1)(m(imgs.double()), target.double())
        with torch.autocast("cpu", dtype=torch.bfloat16):
            pred = m(imgs)
        return ref_loss.ScaleAndShiftInvariantLoss(alpha=0.1)(pred.float(), target)

    run("mae_depth", lambda: rm.ViT_from_MAE(None, False, 1, False, "depth", False, None, 768, 12, 12, "cls"),
        44, ("pos_embed", "decoder_pos_embed"), depth_step)
    # ---- G10's detection trunk at 512^2 (seed 31, inputs 32)
    gg = torch.Generator("cpu").manual_seed(32)
    dimgs = torch.randn(1, 3, 512, 512, generator=gg)
    wgt = torch.randn(1, 1024, 768, generator=gg)

    def det_step(m, mode):
        if mode == "fp64":
            return (m.forward_features(dimgs.double()) * wgt.double()).sum()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            tok = m.forward_features(dimgs)
        return (tok.float() * wgt).sum()

    def det_build():
        m = rm.VisionTransformer_from_Any(False, 0, False, None, True, 512, 768, 12, 12, "cls")
        for p in m.fpn.parameters():
            p.requires_grad_(False)
        return m

    run("t512", det_build, 31, (), det_step)
    np.savez_compressed(os.path.join(HERE, "g17_bf16_bars.npz"), **out)
    print("g17 ok")


def g18_eval_mode():
    """Inference as the reference's eval_*.py / predict_*.py run it: `model.eval()` under `torch.no_grad()` — BatchNorm
    with RUNNING statistics, Dropout off.  The keyed weights start from running_mean = 0 / running_var = 1, so two
    training-mode forwards move the running statistics first (momentum 0.1), then the evaluation batch goes through
    in eval mode: ResNet_from_Any dense + classifier (models.py:106-152), ViT_from_MAE(dense="seg") with the DPT seg
    head's BatchNorm fusion blocks (DPT_decoder.py:461,483-497)."""
    rm = import_reference_models()
    out = {}
    gen = torch.Generator("cpu").manual_seed(81)
    xt = [torch.randn(4, 3, 128, 128, generator=gen) for _ in range(2)]
    xe = torch.randn(4, 3, 128, 128, generator=gen)
    for tag, build, seed in (("resnet_dense", lambda: rm.ResNet_from_Any(None, False, 1, False, "depth"), 82),
                             ("resnet_cls", lambda: rm.ResNet_from_Any(None, True, 6, False, None), 83)):
        m = build()
        shapes, digest = load_keyed(m, seed=seed)
        out[f"{tag}/keys"] = np.array(sorted(shapes)); out[f"{tag}/digest"] = np.array(digest)
        m.train()
        with torch.no_grad():
            for x in xt:
                m(x)
        m.eval()
        with torch.no_grad():
            y = m(xe)
        out[f"{tag}/out"] = y.numpy()
        out[f"{tag}/running_var/layer3.5.bn2"] = m.layer3[5].bn2.running_var.numpy().copy()
        print(f"g18 {tag}: |out| {float(y.double().norm()):.5f}")
    cfg = mae_ref.VIT_B
    imgs_t = [synth.synth_images(2, cfg, seed=84 + i) for i in range(2)]
    imgs_e = synth.synth_images(2, cfg, seed=86)
    m = rm.ViT_from_MAE(None, False, 1, False, "seg", False, None, 768, 12, 12, "cls")
    shapes, digest = load_keyed(m, seed=87, keep=("pos_embed", "decoder_pos_embed"))
    out["mae_seg/keys"] = np.array(sorted(shapes)); out["mae_seg/digest"] = np.array(digest)
    m.train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0   # the two warm-up forwards must be reproducible on the other side
    with torch.no_grad():
        for x in imgs_t:
            m(x)
    m.eval()
    with torch.no_grad():
        y = m(imgs_e)
    out["mae_seg/out_sub"] = y[:, :, ::2, ::2].numpy().copy()
    out["mae_seg/out_norm"] = np.array(float(y.double().norm()))
    np.savez_compressed(os.path.join(HERE, "g18_eval_mode.npz"), **out)
    print(f"g18 ok: seg |out| {float(out['mae_seg/out_norm']):.5f}")


def g15_det_curve(steps=30):
    """the reference's own VisionTransformer_from_Any(det=True) trunk (models.py:155-210 windowed
    blocks, :310-338) at 512 x 512 (1024 tokens, four 256-token windows), B = 1, trained for 30 steps: tokens
    regressed on a fixed random target (the reference trains this trunk inside detectron2, which is not in its
    tree: the loop is zero_grad / forward_features / mean squared error / backward / AdamW(lr 2e-5).step()), two
    seeded batches in rotation.  Weights: the keyed set of G10 (seed 31)."""
    rm = import_reference_models()
    m = rm.VisionTransformer_from_Any(False, 0, False, None, True, 512, 768, 12, 12, "cls")
    shapes, digest = load_keyed(m, seed=31)
    m.train()
    opt = torch.optim.AdamW([p for k, p in m.named_parameters() if not k.startswith("fpn.")], lr=2e-5)
    batches = synth.det_batches()
    losses = []
    for it in range(steps):
        imgs, tgt = batches[it % len(batches)]
        opt.zero_grad()
        loss = ((m.forward_features(imgs) - tgt) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss))
        if it % 5 == 0:
            print(f"  g15 step {it}: {losses[-1]:.6f}", flush=True)
    np.savez_compressed(os.path.join(HERE, "g15_det_curve.npz"), losses=np.array(losses, dtype=np.float64),
                        keys=np.array(sorted(shapes)), digest=np.array(digest), lr=np.array(2e-5),
                        batch=np.array(1), steps=np.array(steps), size=np.array(512), n_batches=np.array(len(batches)))
    print("g15 first/last", losses[0], losses[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-curve", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    torch.set_num_threads(8)
    torch.manual_seed(0)
    ref_mae = import_reference_mae()
    jobs = {
        "g1": lambda: g1_masking(ref_mae), "g2": lambda: g2_patchify(ref_mae), "g3": g3_sincos,
        "g5tiny": lambda: g5_tiny(ref_mae), "g5vitb": lambda: g5_vitb(ref_mae),
        "lr": g_lr_sched, "g6": g6_dpt, "g7": g7_ssi, "g8": g8_moco, "g9": g9_dpt_seg,
        "g3b": g3b_moco_sincos, "g10": g10_det, "g11": g11_vit_api, "g12": g12_resnet_dec,
        "g13": g13_depth_curve, "g14": g14_moco_curve, "g14fp64": g14_moco_fp64, "g15": g15_det_curve,
        "g16": g16_frozen, "g17": g17_bf16_bars, "g18": g18_eval_mode,
        "curve_tiny": lambda: curve(ref_mae, mae_ref.MAEConfig(
            **{**mae_ref.TINY.__dict__, "norm_pix_loss": True}), 8, 100, 1.5e-4,
            "g5_curve_tiny.npz"),
        "curve_vitb": lambda: curve(ref_mae, mae_ref.MAEConfig(
            **{**mae_ref.VIT_B.__dict__, "norm_pix_loss": True}), 8, 100, 1.5e-4,
            "g5_curve_vitb.npz"),
    }
    for name, fn in jobs.items():
        if a.only and name not in a.only.split(","):
            continue
        if a.skip_curve and name == "curve_vitb":
            continue
        fn()


if __name__ == "__main__":
    main()
