"""GPU parity of the finetune zoo against fixtures produced by the
REFERENCE's own `Models/models.py` (tests/golden/make_golden.py: g10 detection backbone + ViTDet_FPN,
g11 ViT_from_MAE / ViT_from_MoCoV3 / VisionTransformer_from_Any heads, dense taps, depth model, g12
ResNet_from_Any decoder).  Weights are oracle.synth.keyed_tensor(key, shape, seed) on both sides,
proven equal by a SHA-256 stored in the fixture.  fp32 engine <= 1e-3 rel on outputs (north_star);
bf16 engine judged on loose output / gradient agreement."""
import numpy as np
import pytest
import torch

from conftest import keyed_weights, load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ssl4gie_amd import _lib
    _lib.load()


def _check_grads(g, prefix, params, tol, names=None, ntol=None):
    gn = dict(zip(g[prefix + "grad_names"].tolist(), g[prefix + "grad_norms"].tolist()))
    ntol = tol if ntol is None else ntol
    for k in (names or gn):
        p = params[k]
        assert p.grad is not None, k
        t = p.grad.detach().float().cpu()
        assert abs(float(t.double().norm()) - gn[k]) <= ntol * gn[k] + 1e-7, (k, float(t.norm()), gn[k])
        if f"{prefix}grad/{k}" in g.files:
            assert rel_err(t, g[f"{prefix}grad/{k}"]) < tol, k
        else:
            ref = torch.from_numpy(g[f"{prefix}gslice/{k}"])
            got = t.reshape(t.shape[0], -1)[:8, :64]
            assert float((got - ref).abs().max()) <= tol * max(float(t.abs().max()), 1e-12), k


DET_NAMES = ("pos_embed", "patch_embed.proj.weight", "patch_embed.proj.bias", "blocks.0.attn.qkv.weight",
             "blocks.0.norm1.weight", "blocks.2.attn.qkv.weight", "blocks.2.attn.proj.bias",
             "blocks.5.mlp.fc2.weight", "blocks.10.attn.proj.weight", "blocks.11.mlp.fc1.weight",
             "blocks.11.mlp.fc1.bias", "norm.weight", "norm.bias")


@pytest.mark.parametrize("prec,tol,gtol", [("fp32", 1e-3, 3e-3), ("bf16", 6e-2, None)])  # bf16 gradients: test_g17_*
def test_g10_det_trunk_512_matches_reference(prec, tol, gtol):
    """reference VisionTransformer_from_Any(det=True, fixed_size=512).forward_features (models.py:
    155-210 windowed blocks 0,1,3,4,6,7,9,10; :310-338): tokens and gradients"""
    from ssl4gie_amd.Models import models
    g = load_golden("g10_det.npz")
    m = models.VisionTransformer_from_Any(False, 0, False, None, True, 1024, 768, 12, 12, "cls")
    keyed_weights(m, 31, g["keys"], g["digest"])
    m.fixed_size = 512  # same module, smaller grid: the trunk does not depend on the pyramid's shapes
    m.patch_embed.img_size = (512, 512)
    m.to(DEV).set_precision(prec)
    gen = torch.Generator("cpu").manual_seed(32)
    imgs = torch.randn(1, 3, 512, 512, generator=gen)
    wgt = torch.randn(1, 1024, 768, generator=gen)
    tok = m.forward_features(imgs.to(DEV))
    assert tok.shape == (1, 1024, 768)
    assert rel_err(tok.detach()[:, ::4], g["t512/tok_sub"]) < tol
    n = float(g["t512/tok_norm"])
    assert abs(float(tok.detach().double().norm()) - n) < tol * n
    (tok * wgt.to(DEV)).sum().backward()
    if gtol is not None:
        _check_grads(g, "t512/", dict(m.named_parameters()), gtol, DET_NAMES)
    # the interpolated position table's gradient is written through the GradSink into the arena
    # (data-parallel buckets and the arena optimizers read the arena, not loose .grad tensors)
    a = m.arena()
    assert all(p.grad is None or p.grad.data_ptr() == a.grad_view(p).data_ptr() for p in m.parameters())


def test_g10_det_backbone_and_pyramid_1024_matches_reference():
    """the whole detection backbone at the reference's hard-coded 1024^2 geometry (N = 4096: streaming
    attention in the 4 global blocks) + ViTDet_FPN (models.py:213-259), B = 1, fp32 engine: the five
    maps torchvision's FasterRCNN consumes, and gradients of trunk and pyramid parameters"""
    from ssl4gie_amd.Models import models
    g = load_golden("g10_det.npz")
    m = models.VisionTransformer_from_Any(False, 0, False, None, True, 1024, 768, 12, 12, "cls")
    keyed_weights(m, 31, g["keys"], g["digest"])
    m.to(DEV).set_precision("fp32")
    imgs = torch.randn(1, 3, 1024, 1024, generator=torch.Generator("cpu").manual_seed(33))
    maps = m(imgs.to(DEV))
    assert list(maps.keys()) == ["0", "1", "2", "3", "pool"]
    loss = 0
    for i, (k, v) in enumerate(maps.items()):
        assert tuple(v.shape) == tuple(g[f"f1024/shape/{k}"]) and v.dtype == torch.float32
        n = float(g[f"f1024/norm/{k}"])
        assert abs(float(v.detach().double().norm()) - n) < 1e-3 * n, k
        c = v.shape[2] // 2
        assert rel_err(v.detach()[:, :8, :16, :16], g[f"f1024/corner/{k}"]) < 2e-3, k
        assert rel_err(v.detach()[:, 100:108, c:c + 4, c:c + 4], g[f"f1024/center/{k}"]) < 2e-3, k
        w = torch.randn(v.shape, generator=torch.Generator("cpu").manual_seed(40 + i))
        loss = loss + (v * w.to(DEV)).sum() / v.numel() ** 0.5
    assert abs(float(loss.detach()) - float(g["f1024/loss"])) < 2e-3 * max(1.0, abs(float(g["f1024/loss"])))
    loss.backward()
    names = DET_NAMES + ("fpn.fpn1.1.weight", "fpn.fpn1.2.weight", "fpn.fpn1.4.bias", "fpn.fpn2.0.bias",
                         "fpn.fpn2.2.weight", "fpn.fpn3.0.weight", "fpn.fpn3.2.bias", "fpn.fpn4.0.bias",
                         "fpn.fpn4.1.weight", "fpn.fpn4.3.weight", "fpn.fpn4.5.weight", "fpn.fpn4.6.weight",
                         "fpn.fpn4.7.bias")
    _check_grads(g, "f1024/", dict(m.named_parameters()), 5e-3, names)


def test_g10_vit_from_mae_det_256_matches_reference():
    """ViT_from_MAE(det=True).forward_encoder (models.py:427-443), one 16 x 16 window"""
    from ssl4gie_amd.Models import models
    g = load_golden("g10_det.npz")
    m = models.ViT_from_MAE(None, False, 0, False, None, True, 1024, 768, 12, 12, "cls")
    keyed_weights(m, 34, g["mae256/keys"], g["mae256/digest"], keep=("pos_embed", "decoder_pos_embed"))
    m.fixed_size = 256
    m.patch_embed.img_size = (256, 256)
    m.to(DEV).set_precision("fp32")
    imgs = torch.randn(2, 3, 256, 256, generator=torch.Generator("cpu").manual_seed(35))
    with torch.no_grad():
        tok = m.forward_encoder(imgs.to(DEV))
    assert rel_err(tok[:, ::2, ::2], g["mae256/tok"]) < 1e-3
    n = float(g["mae256/tok_norm"])
    assert abs(float(tok.double().norm()) - n) < 1e-4 * n


HEAD_NAMES = ("lin_head.weight", "lin_head.bias", "norm.weight", "blocks.11.mlp.fc2.weight",
              "blocks.6.attn.qkv.bias", "blocks.0.attn.qkv.weight", "patch_embed.proj.weight", "cls_token")


@pytest.mark.parametrize("which", ["mae_head", "moco_head", "any_head"])
def test_g11_backbone_heads_match_reference(which):
    """cls / spatial readout + linear head of the three ViT wrappers (models.py:349-354,467-472,
    569-574) with their three different position-embedding conventions"""
    from oracle import mae_ref, synth
    from ssl4gie_amd.Models import models
    g = load_golden("g11_vit_api.npz")
    imgs = synth.synth_images(2, mae_ref.VIT_B, seed=int(g["imgs_seed"]))
    if which == "mae_head":
        m = models.ViT_from_MAE(None, True, 6, False, None, False, None, 768, 12, 12, "cls")
        seed, keep, wseed = 42, ("pos_embed", "decoder_pos_embed"), 43
    elif which == "moco_head":
        m = models.ViT_from_MoCoV3(None, True, 6, False, None, False, None, 768, "cls")
        seed, keep, wseed = 46, ("pos_embed",), 43
    else:
        m = models.VisionTransformer_from_Any(True, 12, False, None, False, None, 768, 12, 12, "cls")
        seed, keep, wseed = 47, (), 48
    keyed_weights(m, seed, g[f"{which}/keys"], g[f"{which}/digest"], keep=keep)
    m.to(DEV).set_precision("fp32")
    y = m(imgs.to(DEV))
    assert rel_err(y.detach(), g[f"{which}/cls"]) < 1e-3
    wy = torch.randn(y.shape, generator=torch.Generator("cpu").manual_seed(wseed))
    (y * wy.to(DEV)).sum().backward()
    names = HEAD_NAMES + (("pos_embed",) if which == "any_head" else ())
    _check_grads(g, f"{which}/", dict(m.named_parameters()), 3e-3, names)
    if which != "any_head":
        assert m.pos_embed.grad is None  # fixed sin-cos tables
        m.out_token = "spatial"
        with torch.no_grad():
            assert rel_err(m(imgs.to(DEV)), g[f"{which}/spatial"]) < 1e-3
    if which == "mae_head":
        with torch.no_grad():
            tok = m.forward_encoder(imgs.to(DEV))
        assert rel_err(tok[:, ::8], g["mae_head/tok_sub"]) < 1e-3


@pytest.mark.parametrize("prec,tol,gtol", [("fp32", 1e-3, 5e-3), ("bf16", 5e-2, None)])  # bf16 gradients: test_g17_*
def test_g11_vit_from_mae_depth_matches_reference(prec, tol, gtol):
    """ViT_from_MAE(dense="depth"): taps after blocks 2/5/8/11 without the final norm (models.py:
    450-456), DPT decoder, the reference's SSI loss, gradients of trunk + decoder; the parameters the
    reference never reaches (`find_unused_parameters=True`) get no gradient here either"""
    from oracle import mae_ref, synth
    from ssl4gie_amd.Models import models
    from ssl4gie_amd.losses import ScaleAndShiftInvariantLoss
    g = load_golden("g11_vit_api.npz")
    imgs = synth.synth_images(2, mae_ref.VIT_B, seed=int(g["imgs_seed"]))
    m = models.ViT_from_MAE(None, False, 1, False, "depth", False, None, 768, 12, 12, "cls")
    keyed_weights(m, 44, g["mae_depth/keys"], g["mae_depth/digest"], keep=("pos_embed", "decoder_pos_embed"))
    m.to(DEV).set_precision(prec)
    with torch.no_grad():
        taps = m.forward_encoder(imgs.to(DEV))
    assert len(taps) == 4
    for i, t in enumerate(taps):
        assert rel_err(t[:, ::16, ::4], g[f"mae_depth/tap_sub/{i}"]) < tol
        n = float(g[f"mae_depth/tap_norm/{i}"])
        assert abs(float(t.double().norm()) - n) < tol * n
    gen = torch.Generator("cpu").manual_seed(45)
    target = torch.rand(2, 1, 224, 224, generator=gen)
    target = torch.where(torch.rand(2, 1, 224, 224, generator=gen) < 0.1, torch.zeros(()), target)
    pred = m(imgs.to(DEV))
    loss = ScaleAndShiftInvariantLoss(alpha=0.1)(pred, target.to(DEV))
    loss.backward()
    assert rel_err(pred.detach()[:, :, ::2, ::2], g["mae_depth/pred_sub"]) < tol
    assert abs(float(loss.detach()) - float(g["mae_depth/loss"])) < tol * float(g["mae_depth/loss"])
    params = dict(m.named_parameters())
    for k in g["mae_depth/no_grad_params"].tolist():
        assert params[k].grad is None or float(params[k].grad.abs().max()) == 0.0, k
    names = ("blocks.11.mlp.fc2.weight", "blocks.8.attn.proj.bias", "blocks.0.attn.qkv.weight",
             "patch_embed.proj.weight", "cls_token", "decoder.act_postprocess12.0.weight",
             "decoder.act_postprocess42.1.weight", "decoder.layer1_rn.weight",
             "decoder.refinenet1.resConfUnit2.conv2.bias", "decoder.refinenet4.out_conv.weight",
             "decoder.output_conv.0.weight", "decoder.output_conv.4.weight")
    if gtol is not None:
        _check_grads(g, "mae_depth/", params, gtol, names)


def _stage_maps(seed, b=2, s=32):
    gen = torch.Generator("cpu").manual_seed(seed)
    return [torch.relu(torch.randn(b, c, s >> i, s >> i, generator=gen)) for i, c in enumerate((256, 512, 1024, 2048))]


def test_g12_resnet_decoder_matches_reference():
    """the reference's own ResNet_from_Any(dense="depth").decode (models.py:16-60,128-135; training-mode
    BatchNorm) on seeded stage maps: prediction, SSI loss, decoder gradients, map gradients, running
    statistics — fp32 engine"""
    from ssl4gie_amd.Models import models
    from ssl4gie_amd.losses import ScaleAndShiftInvariantLoss
    g = load_golden("g12_resnet_dec.npz")
    m = models.ResNet_from_Any(None, False, 1, False, "depth")
    keyed_weights(m, 51, g["keys"], g["digest"])
    m.to(DEV).set_precision("fp32")
    m._prepare()
    maps = [t.permute(0, 2, 3, 1).contiguous().to(DEV).requires_grad_(True) for t in _stage_maps(52)]
    gen = torch.Generator("cpu").manual_seed(53)
    target = torch.rand(2, 1, 128, 128, generator=gen)
    target = torch.where(torch.rand(2, 1, 128, 128, generator=gen) < 0.1, torch.zeros(()), target)
    pred = m.decode(maps)
    assert pred.shape == (2, 1, 128, 128)
    assert rel_err(pred.detach(), g["dec/pred"]) < 1e-3
    loss = ScaleAndShiftInvariantLoss(alpha=0.1)(pred, target.to(DEV))
    assert abs(float(loss.detach()) - float(g["dec/loss"])) < 1e-3 * float(g["dec/loss"])
    loss.backward()
    names = ("output_conv.5.weight", "output_conv.3.weight", "output_conv.1.bias",
             "decoder_levels.2.blocks.2.process.6.weight", "decoder_levels.2.blocks.0.identity.0.weight",
             "decoder_levels.2.blocks.1.process.4.weight", "decoder_levels.1.chan_reduce.0.weight",
             "decoder_levels.1.blocks.1.process.7.bias", "decoder_levels.0.blocks.1.process.3.weight",
             "decoder_levels.0.blocks.0.process.1.weight", "decoder_levels.0.chan_reduce.1.weight")
    # norms to 1 %; elementwise to 4 % of the gradient's scale: a BatchNorm bias gradient is a sum of
    # 512 signed terms here and a single ReLU mask that flips between two fp32 evaluations moves one
    # element by a few % (the fp64 oracle vs the fp32 reference shows the same on the CPU)
    _check_grads(g, "dec/", dict(m.named_parameters()), 4e-2, names, ntol=1e-2)
    for i, t in enumerate(maps):
        gr = t.grad.permute(0, 3, 1, 2).float().cpu()
        n = float(g[f"dec/map_grad_norm/{i}"])
        assert abs(float(gr.double().norm()) - n) < 1e-2 * n, i
        assert float((gr[:, :16, :4, :4] - torch.from_numpy(g[f"dec/map_grad_slice/{i}"])).abs().max()) \
            <= 1e-2 * float(gr.abs().max()), i
    assert rel_err(m.decoder_levels[0].chan_reduce[1].running_mean,
                   g["dec/running_mean/decoder_levels.0.chan_reduce.1"]) < 1e-3
    assert rel_err(m.decoder_levels[2].blocks[2].process[7].running_var,
                   g["dec/running_var/decoder_levels.2.blocks.2.process.7"]) < 1e-3


def test_g12_resnet_whole_model_and_classifier_match_fixture():
    """ResNet_from_Any end to end (trunk generated on oracle/torchvision_restatement.py: the
    torchvision boundary itself stays unpinned) — dense prediction and the pooled + linear-head path"""
    from ssl4gie_amd.Models import models
    g = load_golden("g12_resnet_dec.npz")
    imgs = torch.randn(4, 3, 128, 128, generator=torch.Generator("cpu").manual_seed(54))
    m = models.ResNet_from_Any(None, False, 1, False, "depth")
    keyed_weights(m, 51, g["keys"], g["digest"])
    m.to(DEV).set_precision("fp32")
    with torch.no_grad():
        pred = m(imgs.to(DEV))
    assert rel_err(pred, g["full/pred"]) < 3e-3
    m2 = models.ResNet_from_Any(None, True, 6, False, None)
    keyed_weights(m2, 56, g["cls/keys"], g["cls/digest"])
    m2.to(DEV).set_precision("fp32")
    with torch.no_grad():
        y = m2(imgs.to(DEV))
    assert rel_err(y, g["cls/logits"]) < 3e-3


# ------------------------------------------------------------------ frozen=True (linear probe / frozen finetune)
def _only_these_have_grads(m, prefixes):
    for k, p in m.named_parameters():
        if k.startswith(prefixes):
            assert p.grad is not None, k
        else:
            assert p.grad is None, k


def test_g16_frozen_vit_linear_probe_matches_reference():
    """reference ViT_from_MAE(head=True, frozen=True) (models.py:459-472): the trunk under no_grad (a different
    executor mode: no activation arena, no weight-gradient fork), logits <= 1e-3, only lin_head.* gets gradients"""
    from oracle import mae_ref, synth
    from ssl4gie_amd.Models import models
    g = load_golden("g16_frozen.npz")
    imgs = synth.synth_images(2, mae_ref.VIT_B, seed=int(g["imgs_seed"]))
    m = models.ViT_from_MAE(None, True, 6, True, None, False, None, 768, 12, 12, "cls")
    keyed_weights(m, 72, g["mae_head/keys"], g["mae_head/digest"], keep=("pos_embed", "decoder_pos_embed"))
    m.to(DEV).set_precision("fp32")
    m.train()
    y = m(imgs.to(DEV))
    assert rel_err(y.detach(), g["mae_head/cls"]) < 1e-3
    wy = torch.randn(y.shape, generator=torch.Generator("cpu").manual_seed(73))
    (y * wy.to(DEV)).sum().backward()
    _only_these_have_grads(m, ("lin_head.",))
    _check_grads(g, "mae_head/", dict(m.named_parameters()), 3e-3)
    # bf16 engine: same structure, looser values
    m.zero_grad(set_to_none=True)
    m.set_precision("bf16")
    y = m(imgs.to(DEV))
    assert rel_err(y.detach(), g["mae_head/cls"]) < 5e-2
    (y * wy.to(DEV)).sum().backward()
    _only_these_have_grads(m, ("lin_head.",))


@pytest.mark.parametrize("prec,tol,gtol", [("fp32", 1e-3, 3e-3), ("bf16", 5e-2, None)])
def test_g16_frozen_vit_depth_matches_reference(prec, tol, gtol):
    """reference ViT_from_MAE(dense="depth", frozen=True) (models.py:459-465): frozen trunk, DPT decoder trained with
    the SSI loss — prediction and loss <= 1e-3, decoder gradients <= 3e-3, no trunk gradient"""
    from oracle import mae_ref, synth
    from ssl4gie_amd.Models import models
    from ssl4gie_amd.losses import ScaleAndShiftInvariantLoss
    g = load_golden("g16_frozen.npz")
    imgs = synth.synth_images(2, mae_ref.VIT_B, seed=int(g["imgs_seed"]))
    m = models.ViT_from_MAE(None, False, 1, True, "depth", False, None, 768, 12, 12, "cls")
    keyed_weights(m, 74, g["mae_depth/keys"], g["mae_depth/digest"], keep=("pos_embed", "decoder_pos_embed"))
    m.to(DEV).set_precision(prec)
    m.train()
    gen = torch.Generator("cpu").manual_seed(75)
    target = torch.rand(2, 1, 224, 224, generator=gen)
    target = torch.where(torch.rand(2, 1, 224, 224, generator=gen) < 0.1, torch.zeros(()), target)
    pred = m(imgs.to(DEV))
    loss = ScaleAndShiftInvariantLoss(alpha=0.1)(pred, target.to(DEV))
    loss.backward()
    assert rel_err(pred.detach()[:, :, ::2, ::2], g["mae_depth/pred_sub"]) < tol
    assert abs(float(loss.detach()) - float(g["mae_depth/loss"])) < tol * float(g["mae_depth/loss"])
    with_grad = set(g["mae_depth/with_grad"].tolist())
    for k, p in m.named_parameters():
        if k in with_grad:
            assert p.grad is not None, k
        else:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
        if not k.startswith("decoder."):
            assert p.grad is None, k
    if gtol is not None:
        _check_grads(g, "mae_depth/", dict(m.named_parameters()), gtol)


def test_g16_frozen_resnet_linear_probe_matches_fixture():
    """reference ResNet_from_Any(head=True, frozen=True) (models.py:138-149): the no_grad trunk still runs BatchNorm in
    TRAINING mode (batch statistics, running statistics updated); only lin_head.* gets gradients.  (The trunk below
    the reference class is oracle/torchvision_restatement.py: unpinned at that boundary, as G12.)"""
    from ssl4gie_amd.Models import models
    g = load_golden("g16_frozen.npz")
    m = models.ResNet_from_Any(None, True, 6, True, None)
    keyed_weights(m, 76, g["resnet_head/keys"], g["resnet_head/digest"])
    m.to(DEV).set_precision("fp32")
    m.train()
    imgs = torch.randn(4, 3, 128, 128, generator=torch.Generator("cpu").manual_seed(77))
    y = m(imgs.to(DEV))
    assert rel_err(y.detach(), g["resnet_head/logits"]) < 3e-3
    wy = torch.randn(y.shape, generator=torch.Generator("cpu").manual_seed(78))
    (y * wy.to(DEV)).sum().backward()
    _only_these_have_grads(m, ("lin_head.",))
    _check_grads(g, "resnet_head/", dict(m.named_parameters()), 5e-3)
    from ssl4gie_amd.resnet_engine import flush_batch_counts
    flush_batch_counts(m)
    assert rel_err(m.bn1.running_mean, g["resnet_head/running_mean/bn1"]) < 1e-3
    assert rel_err(m.layer4[2].bn3.running_var, g["resnet_head/running_var/layer4.2.bn3"]) < 3e-3
    assert int(m.bn1.num_batches_tracked) == 1


# ------------------------------------------------------------------ bf16 whole-model gradients, per tensor
def _sample(t, n=2048):
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n]


def _bf16_gate(g, tag, params):
    """per-tensor relative L2 error of the bf16 engine's gradients against the reference's DOUBLE gradients (strided
    samples stored in G17), held to 1.5 x the DISTRIBUTION of the reference's own bf16-autocast error on the same
    samples (tests/golden/make_golden.py g17_bf16_bars): median, 90th percentile, and the worst of the well-sampled
    tensors (>= 1024 elements).  A scalar bias is one signed sum of 1e5 terms — its relative error is a single draw
    of a heavy-tailed quantity in either arithmetic — so the tiny tensors are only held to 3 x the reference's worst."""
    names = g[f"{tag}/names"].tolist()
    ref_err = np.asarray(g[f"{tag}/autocast_err"], dtype=np.float64)
    errs, big = [], []
    for k in names:
        assert params[k].grad is not None, k
        a = torch.from_numpy(g[f"{tag}/sample/{k}"]).double()
        b = _sample(params[k].grad).double().cpu()
        errs.append(float((a - b).norm() / (a.norm() + 1e-300)))
        big.append(params[k].numel() >= 1024)
    errs, big = np.array(errs), np.array(big)
    order = np.argsort(-errs)[:4]
    print(f"{tag}: worst tensors " + ", ".join(f"{names[i]} {errs[i]:.3e} (ref {ref_err[i]:.3e})" for i in order))
    assert np.median(errs) <= 1.5 * np.median(ref_err), (np.median(errs), np.median(ref_err))
    assert np.quantile(errs, 0.9) <= 1.5 * np.quantile(ref_err, 0.9), (np.quantile(errs, 0.9), np.quantile(ref_err, 0.9))
    wb = int(np.argmax(np.where(big, errs, 0)))
    assert errs[wb] <= 1.5 * ref_err[big].max(), (names[wb], errs[wb], ref_err[big].max())
    assert errs.max() <= 3.0 * ref_err.max(), (names[int(errs.argmax())], errs.max(), ref_err.max())
    return errs


def test_g17_bf16_depth_gradients_within_the_references_own_autocast_error():
    """G11's ViT_from_MAE(dense="depth") + SSI loss on the bf16 engine: every gradient tensor against the reference's
    fp64 gradient, gated by the reference's own bf16-autocast error (this loss's gradient is ill-conditioned: the L1
    gradient-matching term of Depth_estimation/Metrics/losses.py:60-77 flips signs under any half-precision rounding
    of the prediction — the reference's autocast is 19-22 % off on every tensor; the fp32 engine is held to 5e-3 by
    test_g11_vit_from_mae_depth_matches_reference)"""
    from oracle import mae_ref, synth
    from ssl4gie_amd.Models import models
    from ssl4gie_amd.losses import ScaleAndShiftInvariantLoss
    g11, g = load_golden("g11_vit_api.npz"), load_golden("g17_bf16_bars.npz")
    imgs = synth.synth_images(2, mae_ref.VIT_B, seed=int(g11["imgs_seed"]))
    m = models.ViT_from_MAE(None, False, 1, False, "depth", False, None, 768, 12, 12, "cls")
    keyed_weights(m, 44, g11["mae_depth/keys"], g11["mae_depth/digest"], keep=("pos_embed", "decoder_pos_embed"))
    m.to(DEV).set_precision("bf16")
    gen = torch.Generator("cpu").manual_seed(45)
    target = torch.rand(2, 1, 224, 224, generator=gen)
    target = torch.where(torch.rand(2, 1, 224, 224, generator=gen) < 0.1, torch.zeros(()), target)
    loss = ScaleAndShiftInvariantLoss(alpha=0.1)(m(imgs.to(DEV)), target.to(DEV))
    loss.backward()
    assert abs(float(loss.detach()) - float(g["mae_depth/loss_fp64"])) < 5e-3 * float(g["mae_depth/loss_fp64"])
    errs = _bf16_gate(g, "mae_depth", dict(m.named_parameters()))
    print(f"bf16 depth gradients vs fp64: median {np.median(errs):.3e} max {errs.max():.3e} "
          f"(reference autocast: {np.median(g['mae_depth/autocast_err']):.3e} / {g['mae_depth/autocast_err'].max():.3e})")


def test_g17_bf16_det_trunk_gradients_within_the_references_own_autocast_error():
    """G10's detection trunk at 512^2 on the bf16 engine: per-tensor gradient error against fp64 within 1.5 x the
    reference's own bf16-autocast error (median 6.3e-3, worst 9.6e-3) — replaces a 25 % whole-model bar"""
    from ssl4gie_amd.Models import models
    g10, g = load_golden("g10_det.npz"), load_golden("g17_bf16_bars.npz")
    m = models.VisionTransformer_from_Any(False, 0, False, None, True, 1024, 768, 12, 12, "cls")
    keyed_weights(m, 31, g10["keys"], g10["digest"])
    m.fixed_size = 512
    m.patch_embed.img_size = (512, 512)
    m.to(DEV).set_precision("bf16")
    gen = torch.Generator("cpu").manual_seed(32)
    imgs = torch.randn(1, 3, 512, 512, generator=gen)
    wgt = torch.randn(1, 1024, 768, generator=gen)
    tok = m.forward_features(imgs.to(DEV))
    (tok.float() * wgt.to(DEV)).sum().backward()
    errs = _bf16_gate(g, "t512", dict(m.named_parameters()))
    print(f"bf16 det-trunk gradients vs fp64: median {np.median(errs):.3e} max {errs.max():.3e} "
          f"(reference autocast: {np.median(g['t512/autocast_err']):.3e} / {g['t512/autocast_err'].max():.3e})")


# ------------------------------------------------------------------ model.eval() inference (eval_*.py / predict_*.py)
@pytest.mark.parametrize("prec,tol", [("fp32", 2e-3), ("bf16", 6e-2)])
@pytest.mark.parametrize("tag", ["resnet_dense", "resnet_cls"])
def test_g18_resnet_eval_mode_matches_reference(tag, prec, tol):
    """reference ResNet_from_Any in eval mode under no_grad (BatchNorm with running statistics), after two
    training-mode forwards have moved those statistics on both sides (models.py:106-152)"""
    from ssl4gie_amd.Models import models
    from ssl4gie_amd.resnet_engine import flush_batch_counts
    g = load_golden("g18_eval_mode.npz")
    gen = torch.Generator("cpu").manual_seed(81)
    xt = [torch.randn(4, 3, 128, 128, generator=gen) for _ in range(2)]
    xe = torch.randn(4, 3, 128, 128, generator=gen)
    if tag == "resnet_dense":
        m, seed = models.ResNet_from_Any(None, False, 1, False, "depth"), 82
    else:
        m, seed = models.ResNet_from_Any(None, True, 6, False, None), 83
    keyed_weights(m, seed, g[f"{tag}/keys"], g[f"{tag}/digest"])
    m.to(DEV).set_precision(prec)
    m.train()
    with torch.no_grad():
        for x in xt:
            m(x.to(DEV))
    m.eval()
    with torch.no_grad():
        y = m(xe.to(DEV))
    assert y.dtype == torch.float32 and tuple(y.shape) == tuple(g[f"{tag}/out"].shape)
    assert rel_err(y, g[f"{tag}/out"]) < tol
    flush_batch_counts(m)
    assert rel_err(m.layer3[5].bn2.running_var, g[f"{tag}/running_var/layer3.5.bn2"]) < (2e-3 if prec == "fp32" else 5e-2)
    assert int(m.bn1.num_batches_tracked) == 2


@pytest.mark.parametrize("prec,tol", [("fp32", 2e-3), ("bf16", 6e-2)])
def test_g18_vit_seg_eval_mode_matches_reference(prec, tol):
    """reference ViT_from_MAE(dense="seg") in eval mode: the DPT seg head's BatchNorm fusion blocks on running
    statistics, Dropout off (DPT_decoder.py:461,483-497)"""
    from oracle import mae_ref, synth
    from ssl4gie_amd.Models import models
    g = load_golden("g18_eval_mode.npz")
    cfg = mae_ref.VIT_B
    imgs_t = [synth.synth_images(2, cfg, seed=84 + i) for i in range(2)]
    imgs_e = synth.synth_images(2, cfg, seed=86)
    m = models.ViT_from_MAE(None, False, 1, False, "seg", False, None, 768, 12, 12, "cls")
    keyed_weights(m, 87, g["mae_seg/keys"], g["mae_seg/digest"], keep=("pos_embed", "decoder_pos_embed"))
    m.to(DEV).set_precision(prec)
    m.train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    with torch.no_grad():
        for x in imgs_t:
            m(x.to(DEV))
    m.eval()
    with torch.no_grad():
        y = m(imgs_e.to(DEV))
    assert rel_err(y[:, :, ::2, ::2], g["mae_seg/out_sub"]) < tol
    n = float(g["mae_seg/out_norm"])
    assert abs(float(y.double().norm()) - n) < tol * n
