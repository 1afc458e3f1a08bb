"""CPU: the oracle restatements of the finetune zoo (oracle/det_ref.py, resnet_ref.py, mae_ref.vit_trunk,
sincos_2d_moco) are pinned against fixtures generated from the REFERENCE's own `Models/models.py` and
`Models/moco_v3/vits.py` (tests/golden/make_golden.py: g3b, g10, g11, g12), and the build's modules
carry the reference's state_dict schema.  No GPU, no /root/reference at run time."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err, keyed_weights
from oracle import det_ref, dpt_ref, mae_ref, resnet_ref, synth


def _check_grads(g, prefix, grads, tol, names=None):
    """grads: name -> tensor; compared with the fixture's norms / full small grads / slices"""
    gn = dict(zip(g[prefix + "grad_names"].tolist(), g[prefix + "grad_norms"].tolist()))
    for k in (names or gn):
        assert k in grads and grads[k] is not None, k
        t = grads[k].detach().float().cpu()
        assert abs(float(t.double().norm()) - gn[k]) <= tol * gn[k] + 1e-7, (k, float(t.norm()), gn[k])
        if f"{prefix}grad/{k}" in g.files:
            assert rel_err(t, g[f"{prefix}grad/{k}"]) < tol, k
        else:
            ref = torch.from_numpy(g[f"{prefix}gslice/{k}"])
            got = t.reshape(t.shape[0], -1)[:8, :64]
            # slice error relative to the scale of the whole gradient
            assert float((got - ref).abs().max()) <= tol * max(float(t.abs().max()), 1e-12), k


def test_g3b_moco_sincos_table_is_the_reference_table():
    g = load_golden("g3_sincos.npz")
    from ssl4gie_amd.Models.models import moco_sincos_pos_embed
    for d, grid, key in ((768, 14, "moco_768"), (384, 14, "moco_384"), (192, 4, "moco_192_g4")):
        ref = g[key]
        assert ref.shape == (1, grid * grid + 1, d)
        assert np.array_equal(mae_ref.sincos_2d_moco(d, grid).numpy(), ref)       # oracle
        assert np.array_equal(moco_sincos_pos_embed(d, (grid, grid)).numpy(), ref)  # product host table
    assert not np.array_equal(g["moco_768"][0], g["mae_768"])  # the two layouts differ (w-major vs h-major)


def test_g10_det_trunk_oracle_matches_reference_fixture():
    g = load_golden("g10_det.npz")
    shapes = {k: None for k in g["keys"].tolist()}
    assert "cls_token" not in shapes and "fpn.fpn4.7.bias" in shapes and len(shapes) == 189
    from ssl4gie_amd.Models import models
    m = models.VisionTransformer_from_Any(False, 0, False, None, True, 1024, 768, 12, 12, "cls")
    own = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert sorted(own) == sorted(shapes)  # the build's module carries the reference's schema
    sd = synth.keyed_state_dict(own, 31)
    assert synth.state_dict_digest(sd) == str(g["digest"])
    gen = torch.Generator("cpu").manual_seed(32)
    imgs = torch.randn(1, 3, 512, 512, generator=gen)
    wgt = torch.randn(1, 1024, 768, generator=gen)
    names = ("pos_embed", "patch_embed.proj.weight", "blocks.0.attn.qkv.weight", "blocks.2.attn.qkv.weight",
             "blocks.10.attn.proj.weight", "blocks.11.mlp.fc1.weight", "norm.weight", "norm.bias")
    sdo = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    tok = det_ref.det_trunk(sdo, imgs, 512)
    assert rel_err(tok.detach()[:, ::4], g["t512/tok_sub"]) < 1e-3
    assert abs(float(tok.detach().double().norm()) - float(g["t512/tok_norm"])) < 1e-4 * float(g["t512/tok_norm"])
    (tok * wgt).sum().backward()
    _check_grads(g, "t512/", {k: sdo[k].grad for k in names}, 2e-3, names)


def test_g10_det_mae_trunk_and_full_pyramid_oracle():
    g = load_golden("g10_det.npz")
    from ssl4gie_amd.Models import models
    m = models.ViT_from_MAE(None, False, 0, False, None, True, 1024, 768, 12, 12, "cls")
    own = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert sorted(own) == g["mae256/keys"].tolist()
    keep = ("pos_embed", "decoder_pos_embed")
    sd = synth.keyed_state_dict(own, 34, keep=keep)
    sd.update({k: m.state_dict()[k].clone() for k in keep})  # the fixed tables the module built
    assert synth.state_dict_digest(sd) == str(g["mae256/digest"])
    imgs = torch.randn(2, 3, 256, 256, generator=torch.Generator("cpu").manual_seed(35))
    with torch.no_grad():
        tok = det_ref.det_trunk(sd, imgs, 256)
    assert rel_err(tok[:, ::2, ::2], g["mae256/tok"]) < 1e-3
    # whole backbone + pyramid at the reference's hard-coded 1024^2 geometry (forward only on the CPU)
    m2 = models.VisionTransformer_from_Any(False, 0, False, None, True, 1024, 768, 12, 12, "cls")
    sd = synth.keyed_state_dict({k: tuple(v.shape) for k, v in m2.state_dict().items()}, 31)
    imgs = torch.randn(1, 3, 1024, 1024, generator=torch.Generator("cpu").manual_seed(33))
    with torch.no_grad():
        maps = det_ref.fpn(sd, det_ref.det_trunk(sd, imgs, 1024))
    for k, v in maps.items():
        assert tuple(v.shape) == tuple(g[f"f1024/shape/{k}"])
        n = float(g[f"f1024/norm/{k}"])
        assert abs(float(v.double().norm()) - n) < 1e-3 * n, k
        c = v.shape[2] // 2
        assert rel_err(v[:, :8, :16, :16], g[f"f1024/corner/{k}"]) < 2e-3, k
        assert rel_err(v[:, 100:108, c:c + 4, c:c + 4], g[f"f1024/center/{k}"]) < 2e-3, k


def test_g11_vit_api_oracle_matches_reference_fixture():
    g = load_golden("g11_vit_api.npz")
    cfg = mae_ref.VIT_B
    imgs = synth.synth_images(2, cfg, seed=int(g["imgs_seed"]))
    from ssl4gie_amd.Models import models
    # ViT_from_MAE + head: schema, cls / spatial logits, head gradients
    m = models.ViT_from_MAE(None, True, 6, False, None, False, None, 768, 12, 12, "cls")
    sd = keyed_weights(m, 42, g["mae_head/keys"], g["mae_head/digest"], keep=("pos_embed", "decoder_pos_embed"))
    with torch.no_grad():
        tok = mae_ref.vit_trunk(sd, cfg, imgs, False)
    assert rel_err(tok[:, ::8], g["mae_head/tok_sub"]) < 1e-3
    lin = lambda t: t @ sd["lin_head.weight"].T + sd["lin_head.bias"]
    assert rel_err(lin(tok[:, 0]), g["mae_head/cls"]) < 1e-3
    assert rel_err(lin(tok[:, 1:].mean(1)), g["mae_head/spatial"]) < 1e-3
    # ViT_from_MoCoV3: cat cls, THEN add the fixed table whose cls row is zero (timm _pos_embed)
    m = models.ViT_from_MoCoV3(None, True, 6, False, None, False, None, 768, "cls")
    sd = keyed_weights(m, 46, g["moco_head/keys"], g["moco_head/digest"], keep=("pos_embed",))
    with torch.no_grad():
        tok = mae_ref.vit_trunk(sd, cfg, imgs, False)  # cls + pos[0] with pos[0] = 0: same arithmetic
    lin = lambda t: t @ sd["lin_head.weight"].T + sd["lin_head.bias"]
    assert rel_err(lin(tok[:, 0]), g["moco_head/cls"]) < 1e-3
    assert rel_err(lin(tok[:, 1:].mean(1)), g["moco_head/spatial"]) < 1e-3
    # VisionTransformer_from_Any: learned table incl. the cls row
    m = models.VisionTransformer_from_Any(True, 12, False, None, False, None, 768, 12, 12, "cls")
    sd = keyed_weights(m, 47, g["any_head/keys"], g["any_head/digest"])
    with torch.no_grad():
        tok = mae_ref.vit_trunk(sd, cfg, imgs, False)
    assert rel_err(tok[:, 0] @ sd["lin_head.weight"].T + sd["lin_head.bias"], g["any_head/cls"]) < 1e-3


def test_g11_dense_taps_and_depth_oracle():
    g = load_golden("g11_vit_api.npz")
    cfg = mae_ref.VIT_B
    imgs = synth.synth_images(2, cfg, seed=int(g["imgs_seed"]))
    from ssl4gie_amd.Models import models
    m = models.ViT_from_MAE(None, False, 1, False, "depth", False, None, 768, 12, 12, "cls")
    sd = keyed_weights(m, 44, g["mae_depth/keys"], g["mae_depth/digest"], keep=("pos_embed", "decoder_pos_embed"))
    assert g["mae_depth/no_grad_params"].tolist() == sorted(
        ["norm.weight", "norm.bias"] + [f"decoder.refinenet4.resConfUnit1.conv{i}.{p}" for i in (1, 2)
                                         for p in ("weight", "bias")])  # never-ready params
    with torch.no_grad():
        taps = mae_ref.vit_trunk(sd, cfg, imgs, True)
        pred = dpt_ref.dpt_forward({k[8:]: v for k, v in sd.items() if k.startswith("decoder.")}, taps)
    for i, t in enumerate(taps):
        assert rel_err(t[:, ::16, ::4], g[f"mae_depth/tap_sub/{i}"]) < 1e-3
        n = float(g[f"mae_depth/tap_norm/{i}"])
        assert abs(float(t.double().norm()) - n) < 1e-4 * n
    assert rel_err(pred[:, :, ::2, ::2], g["mae_depth/pred_sub"]) < 1e-3
    gen = torch.Generator("cpu").manual_seed(45)
    target = torch.rand(2, 1, 224, 224, generator=gen)
    target = torch.where(torch.rand(2, 1, 224, 224, generator=gen) < 0.1, torch.zeros(()), target)
    loss = dpt_ref.ssi_loss(pred, target, alpha=0.1)
    assert abs(float(loss) - float(g["mae_depth/loss"])) < 1e-4 * float(g["mae_depth/loss"])


def _stage_maps(seed, b=2, s=32):
    gen = torch.Generator("cpu").manual_seed(seed)
    return [torch.relu(torch.randn(b, c, s >> i, s >> i, generator=gen)) for i, c in enumerate((256, 512, 1024, 2048))]


def test_g12_resnet_decoder_oracle_matches_reference_fixture():
    g = load_golden("g12_resnet_dec.npz")
    from ssl4gie_amd.Models import models
    m = models.ResNet_from_Any(None, False, 1, False, "depth")
    sd = keyed_weights(m, 51, g["keys"], g["digest"])
    assert len(sd) == 555
    names = ("output_conv.5.weight", "output_conv.3.weight", "output_conv.1.bias",
             "decoder_levels.2.blocks.2.process.6.weight", "decoder_levels.2.blocks.0.identity.0.weight",
             "decoder_levels.1.chan_reduce.0.weight", "decoder_levels.0.blocks.1.process.3.weight",
             "decoder_levels.0.blocks.0.process.1.weight", "decoder_levels.1.blocks.1.process.7.bias")
    sdo = {k: (v.double().requires_grad_(True) if k in names else (v.double() if v.is_floating_point() else v))
           for k, v in sd.items()}
    maps = [t.double().requires_grad_(True) for t in _stage_maps(52)]
    pred = resnet_ref.decode(sdo, maps)
    assert rel_err(pred.detach().float(), g["dec/pred"]) < 1e-3
    gen = torch.Generator("cpu").manual_seed(53)
    target = torch.rand(2, 1, 128, 128, generator=gen)
    target = torch.where(torch.rand(2, 1, 128, 128, generator=gen) < 0.1, torch.zeros(()), target)
    loss = dpt_ref.ssi_loss(pred, target.double(), alpha=0.1)
    assert abs(float(loss) - float(g["dec/loss"])) < 1e-4 * float(g["dec/loss"])
    loss.backward()
    _check_grads(g, "dec/", {k: sdo[k].grad for k in names}, 5e-3, names)
    for i, t in enumerate(maps):
        n = float(g[f"dec/map_grad_norm/{i}"])
        assert abs(float(t.grad.norm()) - n) < 5e-3 * n
        assert float((t.grad[:, :16, :4, :4].float() - torch.from_numpy(g[f"dec/map_grad_slice/{i}"])).abs().max()) \
            <= 5e-3 * float(t.grad.abs().max())
    # whole model (trunk unpinned at the torchvision boundary) and the classification path
    imgs = torch.randn(4, 3, 128, 128, generator=torch.Generator("cpu").manual_seed(54))
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    with torch.no_grad():
        assert rel_err(resnet_ref.resnet50_dense(sd64, imgs.double()).float(), g["full/pred"]) < 2e-3
    m2 = models.ResNet_from_Any(None, True, 6, False, None)
    sd2 = keyed_weights(m2, 56, g["cls/keys"], g["cls/digest"])
    with torch.no_grad():
        feat = resnet_ref.resnet50_pooled({k: (v.double() if v.is_floating_point() else v) for k, v in sd2.items()},
                                          imgs.double())
        logits = feat @ sd2["lin_head.weight"].double().T + sd2["lin_head.bias"].double()
    assert rel_err(logits.float(), g["cls/logits"]) < 2e-3


def test_g13_depth_curve_first_step_oracle():
    """the first value of the reference's depth curve (tests/golden/make_golden.py g13) from the oracle: ViT-B
    trunk with taps + DPT decoder + SSI loss on the same keyed weights and the first seeded batch"""
    g = load_golden("g13_depth_curve.npz")
    cfg = mae_ref.VIT_B
    shapes = None
    # key -> shape of the reference's ViT_from_MAE(dense="depth") state_dict: the build's module has the same schema
    from ssl4gie_amd.Models import models
    m = models.ViT_from_MAE(None, False, 1, False, "depth", False, None, 768, 12, 12, "cls")
    sd = keyed_weights(m, 44, g["keys"], g["digest"], keep=("pos_embed", "decoder_pos_embed"))
    imgs, target = synth.depth_batches()[0]
    with torch.no_grad():
        taps = mae_ref.vit_trunk(sd, cfg, imgs, True)
        pred = dpt_ref.dpt_forward({k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}, taps)
        loss = dpt_ref.ssi_loss(pred, target, alpha=0.1)
    assert abs(float(loss) - float(g["losses"][0])) < 1e-4 * float(g["losses"][0])


def test_g14_moco_curve_first_step_oracle_fp64():
    """the first value of the reference's MoCo-R50 curve (g14) from the fp64 oracle composition (ResNet50 trunk,
    MLP heads, InfoNCE) — the forward is well-conditioned (1.7e-6), unlike its gradients (test_gpu_curves.py)"""
    from oracle import moco_ref
    g = load_golden("g14_moco_curve.npz")
    from functools import partial
    from ssl4gie_amd.Models.moco_v3.moco import builder
    from ssl4gie_amd.Models.resnet import resnet50
    torch.manual_seed(0)
    m = builder.MoCo_ResNet(partial(resnet50, zero_init_residual=True), 256, 1024, 1.0)
    own = m.state_dict()
    assert sorted(own) == sorted(g["keys"].tolist())
    sd32 = synth.keyed_state_dict({k: tuple(v.shape) for k, v in own.items()}, 61)
    for k in list(sd32):
        kb = "base_encoder." + k[len("momentum_encoder."):]
        if k.startswith("momentum_encoder.") and "running" not in k and "num_batches" not in k:
            sd32[k] = sd32[kb].clone()
    assert synth.state_dict_digest(sd32) == str(g["digest"])
    sd = {k: (v.double().requires_grad_(k.startswith(("base_encoder.", "predictor.")) and "running" not in k)
              if v.is_floating_point() else v) for k, v in sd32.items()}
    x1, x2 = synth.moco_views(b=16, size=128)[0]
    x1, x2 = x1.double(), x2.double()

    def enc(prefix, x):
        sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
        return moco_ref.mlp_forward(sub, "fc.", resnet_ref.resnet50_pooled(sub, x))

    pred = {k[len("predictor."):]: v for k, v in sd.items() if k.startswith("predictor.")}
    q1 = moco_ref.mlp_forward(pred, "", enc("base_encoder.", x1))
    q2 = moco_ref.mlp_forward(pred, "", enc("base_encoder.", x2))
    with torch.no_grad():
        k1, k2 = enc("momentum_encoder.", x1), enc("momentum_encoder.", x2)
    loss = moco_ref.contrastive_loss(q1, k2, 1.0) + moco_ref.contrastive_loss(q2, k1, 1.0)
    assert abs(float(loss) - float(g["losses"][0])) < 1e-5 * float(g["losses"][0])
    # ... and against the reference classes THEMSELVES evaluated in fp64 (g14_moco_fp64.npz, the ground truth of the
    # MoCo parity gates in test_gpu_curves.py): loss and every step-0 gradient, double against double
    g64 = load_golden("g14_moco_fp64.npz")
    assert str(g64["digest"]) == str(g["digest"])
    assert abs(float(loss) - float(g64["losses"][0])) < 1e-10 * float(g64["losses"][0])
    loss.backward()
    worst = 0.0
    for k in g64["step0/grad_names"].tolist():
        t = sd[k].grad
        if f"step0/grad/{k}" in g64.files:
            r = torch.from_numpy(g64[f"step0/grad/{k}"])
        else:
            r, t = torch.from_numpy(g64[f"step0/gslice/{k}"]), t.reshape(t.shape[0], -1)[:8, :64]
        e = float((t - r).norm() / (r.norm() + 1e-300))
        worst = max(worst, e)
        assert e < 1e-7, (k, e)
    print(f"fp64 oracle vs fp64 reference, step-0 gradients: worst relative L2 error {worst:.2e}")


def test_g15_det_curve_first_step_oracle():
    """the first value of the reference's detection-trunk curve (g15: VisionTransformer_from_Any(det=True) at
    512 x 512, tokens regressed on a fixed target) from the oracle's trunk on the same keyed weights and batch"""
    from oracle import det_ref
    from ssl4gie_amd.Models import models
    g = load_golden("g15_det_curve.npz")
    m = models.VisionTransformer_from_Any(False, 0, False, None, True, 1024, 768, 12, 12, "cls")
    sd = keyed_weights(m, 31, g["keys"], g["digest"])
    imgs, tgt = synth.det_batches()[0]
    with torch.no_grad():
        tok = det_ref.det_trunk(sd, imgs, 512)
        loss = ((tok - tgt) ** 2).mean()
    assert abs(float(loss) - float(g["losses"][0])) < 1e-4 * float(g["losses"][0])
