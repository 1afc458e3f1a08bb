"""CPU: the oracle (oracle/mae_ref.py) is pinned against fixtures generated from the REFERENCE's own
classes (tests/golden/make_golden.py).  No GPU, no /root/reference at run time."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err
from oracle import mae_ref, synth


def test_g1_masking_bit_exact():
    g = load_golden("g1_masking.npz")
    ids_shuffle, ids_restore, ids_keep, mask = mae_ref.masking_from_noise(g["noise"], 0.75)
    assert np.array_equal(ids_shuffle, g["ids_shuffle"])
    assert np.array_equal(ids_restore, g["ids_restore"])
    assert np.array_equal(mask, g["mask"])
    assert (mask.sum(1) == 147).all()
    x = torch.from_numpy(g["x"])
    xm = torch.gather(x, 1, torch.from_numpy(ids_keep)[:, :, None].expand(-1, -1, x.shape[2]))
    assert np.array_equal(xm.numpy(), g["x_masked"])


def test_g1_ties_are_stable():
    g = load_golden("g1_masking.npz")
    s, r, _, m = mae_ref.masking_from_noise(g["tie_noise"], 0.75)
    assert np.array_equal(s, g["tie_ids_shuffle"]) and np.array_equal(r, g["tie_ids_restore"])
    assert np.array_equal(s[1], np.arange(196))  # all-equal row keeps index order
    assert np.array_equal(m, g["tie_mask"])


def test_g2_patchify_exact():
    g = load_golden("g2_patchify.npz")
    imgs = torch.from_numpy(g["imgs"])
    p = mae_ref.patchify(imgs, 16)
    assert np.array_equal(p.numpy(), g["patches"])
    assert torch.equal(mae_ref.unpatchify(p, 16), imgs)
    big = synth.synth_images(1, mae_ref.VIT_B, seed=3)
    assert np.array_equal(mae_ref.patchify(big, 16)[0, [0, 13, 14, 195]].numpy(), g["big_rows"])


def test_g3_sincos_tables():
    g = load_golden("g3_sincos.npz")
    for d in (768, 512, 192, 128):
        t = mae_ref.sincos_2d(d, 14).astype(np.float32)
        np.testing.assert_allclose(t, g[f"mae_{d}"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(mae_ref.sincos_2d(192, 4).astype(np.float32), g["mae_192_g4"],
                               rtol=0, atol=1e-6)


@pytest.mark.parametrize("tag,npl", [("raw", False), ("npl", True)])
def test_g5_tiny_forward_and_all_grads(tag, npl):
    g = load_golden("g5_mae_tiny.npz")
    cfg = mae_ref.MAEConfig(**{**mae_ref.TINY.__dict__, "norm_pix_loss": npl})
    sd = synth.mae_state_dict(cfg, seed=1)
    assert synth.state_dict_digest(sd) == str(g[f"{tag}/digest"])
    sd = {k: v.clone().requires_grad_("pos_embed" not in k) for k, v in sd.items()}
    imgs = synth.synth_images(4, cfg, seed=1)
    noise = synth.synth_noise(4, cfg.num_patches, seed=1)
    loss, pred, mask, _ = mae_ref.mae_forward(sd, cfg, imgs, noise)
    assert abs(float(loss) - float(g[f"{tag}/loss"])) <= 1e-5 * abs(float(g[f"{tag}/loss"]))
    assert rel_err(pred.detach(), g[f"{tag}/pred"]) < 1e-4
    assert np.array_equal(mask.numpy(), g[f"{tag}/mask"])
    loss.backward()
    n = 0
    for k, v in sd.items():
        key = f"{tag}/grad/{k}"
        if key in g.files:
            assert rel_err(v.grad, g[key]) < 2e-4, k
            n += 1
    assert n >= 60


def test_g5_vitb_forward():
    g = load_golden("g5_mae_vitb.npz")
    assert int(g["n_tensors"]) == 254 and int(g["n_params"]) == 111907840
    cfg = mae_ref.MAEConfig(**{**mae_ref.VIT_B.__dict__, "norm_pix_loss": True})
    sd = synth.mae_state_dict(cfg, seed=0)
    assert synth.state_dict_digest(sd) == str(g["digest"])
    imgs = synth.synth_images(2, cfg, seed=0)
    noise = synth.synth_noise(2, 196, seed=0)
    with torch.no_grad():
        loss, pred, mask, _ = mae_ref.mae_forward(sd, cfg, imgs, noise)
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * float(g["loss"])
    assert rel_err(pred, g["pred"]) < 1e-4
    assert np.array_equal(mask.numpy(), g["mask"])


def test_g5_curve_tiny_oracle_reproduces_reference_training():
    """AdamW(b=(0.9,0.95), wd 0.05 on >1-D non-bias params) x 100 steps; main_pretrain.py:179-180."""
    g = load_golden("g5_curve_tiny.npz")
    cfg = mae_ref.MAEConfig(**{**mae_ref.TINY.__dict__, "norm_pix_loss": True})
    sd = synth.mae_state_dict(cfg, seed=0)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if "pos_embed" not in k}
    full = dict(sd)
    full.update(params)
    decay = [v for k, v in params.items() if v.ndim > 1 and not k.endswith(".bias")]
    no_decay = [v for k, v in params.items() if not (v.ndim > 1 and not k.endswith(".bias"))]
    opt = torch.optim.AdamW([{"params": no_decay, "weight_decay": 0.0},
                             {"params": decay, "weight_decay": 0.05}], lr=float(g["lr"]),
                            betas=(0.9, 0.95))
    b = int(g["batch"])
    steps = 30  # the first 30 of the 100 reference steps keep the CPU suite short
    for it in range(steps):
        imgs = synth.synth_images(b, cfg, seed=it % 4)
        noise = synth.synth_noise(b, cfg.num_patches, seed=100 + it)
        opt.zero_grad(set_to_none=True)
        loss, _, _, _ = mae_ref.mae_forward(full, cfg, imgs, noise)
        loss.backward()
        opt.step()
        assert abs(float(loss) - g["losses"][it]) < 1e-3 * g["losses"][it], it


def test_lr_schedule_mirror():
    import types
    from ssl4gie_amd.Models.mae.util import lr_sched
    g = load_golden("g_lr_sched.npz")
    args = types.SimpleNamespace(lr=float(g["base_lr"]), min_lr=float(g["min_lr"]),
                                 warmup_epochs=int(g["warmup_epochs"]), epochs=int(g["total_epochs"]))

    class Opt:
        param_groups = [{"lr": 0.0}, {"lr": 0.0, "lr_scale": 0.5}]

    for e, lr, lrs in zip(g["epochs"], g["lr"], g["lr_scaled"]):
        lr_sched.adjust_learning_rate(Opt, float(e), args)
        assert abs(Opt.param_groups[0]["lr"] - lr) <= 1e-12 + 1e-9 * abs(lr)
        assert abs(Opt.param_groups[1]["lr"] - lrs) <= 1e-12 + 1e-9 * abs(lrs)


# ------------------------------------------------------------------ DPT depth decoder + SSI loss
def _dpt_inputs(seed, b=2):
    g = torch.Generator("cpu").manual_seed(seed)
    acts = [torch.randn(b, 197, 768, generator=g) for _ in range(4)]
    target = torch.rand(b, 1, 224, 224, generator=g)
    target = torch.where(torch.rand(b, 1, 224, 224, generator=g) < 0.1, torch.zeros(()), target)
    return acts, target


def test_g6_dpt_oracle_matches_reference_fixture():
    """oracle/dpt_ref.py against outputs of the reference's own DPT_decoder + SSI loss"""
    from oracle import dpt_ref
    g = load_golden("g6_dpt_depth.npz")
    sd = {k: v.requires_grad_(True) for k, v in dpt_ref.dpt_state_dict(int(g["seed_weights"])).items()}
    acts, target = _dpt_inputs(int(g["seed_inputs"]))
    acts = [a.requires_grad_(True) for a in acts]
    out, mid = dpt_ref.dpt_forward(sd, acts, return_all=True)
    loss = dpt_ref.ssi_loss(out, target, alpha=0.1)
    loss.backward()
    assert rel_err(out, g["out"]) < 1e-5
    assert rel_err(mid["layer_4"], g["layer_4"]) < 1e-5
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    no_grad = set(str(n) for n in g["no_grad_params"])
    assert no_grad == {f"refinenet4.resConfUnit1.conv{c}.{t}" for c in (1, 2) for t in ("weight", "bias")}
    for name, ref_norm in zip(g["grad_names"], g["grad_norms"]):
        got = float(sd[str(name)].grad.norm())
        assert abs(got - ref_norm) <= 1e-3 * max(ref_norm, 1e-12), name
    for i in range(4):
        assert rel_err(acts[i].grad[:, :4, :64], g[f"act_grad_slice/{i}"]) < 1e-3  # fp32 conv reduction order


def test_g9_dpt_seg_oracle_matches_reference_fixture():
    """oracle/dpt_ref.py seg restatement against outputs of the reference's own
    DPT_decoder(dense="seg") + SoftDiceLoss (training-mode BatchNorm, Dropout.p = 0)"""
    from oracle import dpt_ref
    g = load_golden("g9_dpt_seg.npz")
    sd = {k: v.requires_grad_(True) for k, v in dpt_ref.seg_state_dict(int(g["seed_weights"])).items()}
    acts, _ = _dpt_inputs(int(g["seed_inputs"]))
    acts = [a.requires_grad_(True) for a in acts]
    target = torch.from_numpy(g["target"]).float()
    out = dpt_ref.seg_forward(sd, acts)
    loss = dpt_ref.soft_dice_loss(out, target)
    loss.backward()
    assert rel_err(out, g["out"]) < 1e-4
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    for name, ref_norm in zip(g["grad_names"], g["grad_norms"]):
        got = float(sd[str(name)].grad.norm())
        assert abs(got - ref_norm) <= 5e-3 * max(ref_norm, 1e-12), name
    for i in range(4):
        assert rel_err(acts[i].grad[:, :4, :64], g[f"act_grad_slice/{i}"]) < 5e-3


def test_dpt_seg_module_schema_and_dice_loss_mirror():
    """state_dict keys of the engine's DPT_decoder(dense="seg") == the reference class's (incl.
    BatchNorm buffers); host-side SoftDiceLoss mirror == oracle"""
    from oracle import dpt_ref
    from ssl4gie_amd.Models.DPT_decoder import DPT_decoder
    from ssl4gie_amd.losses import SoftDiceLoss
    g = load_golden("g9_dpt_seg.npz")
    m = DPT_decoder(num_classes=1, dense="seg")
    assert sorted(m.state_dict().keys()) == [str(k) for k in g["state_dict_keys"]]
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()
              if "running" not in k and "num_batches" not in k}
    assert shapes == dpt_ref.seg_param_shapes()
    logits = torch.randn(3, 1, 32, 32, generator=torch.Generator().manual_seed(1)).requires_grad_(True)
    tgt = (torch.rand(3, 1, 32, 32, generator=torch.Generator().manual_seed(2)) < 0.4).float()
    a = SoftDiceLoss()(logits, tgt)
    b = dpt_ref.soft_dice_loss(logits.detach(), tgt)
    assert abs(float(a) - float(b)) < 1e-7


def test_g7_ssi_loss_oracle_and_host_mirror():
    from oracle import dpt_ref
    from ssl4gie_amd.losses import ScaleAndShiftInvariantLoss
    g = load_golden("g7_ssi_loss.npz")
    target = torch.from_numpy(g["target"])
    for fn in (lambda p: dpt_ref.ssi_loss(p, target, alpha=0.1),
               lambda p: ScaleAndShiftInvariantLoss(alpha=0.1)(p, target)):
        pred = torch.from_numpy(g["pred"]).clone().requires_grad_(True)
        loss = fn(pred)
        loss.backward()
        assert abs(float(loss) - float(g["loss"])) < 1e-6 * abs(float(g["loss"]))
        assert rel_err(pred.grad, g["grad"]) < 1e-5


def test_dpt_module_schema_matches_reference():
    """state_dict names / shapes of the engine's DPT_decoder == the reference class's (64 tensors)"""
    from oracle import dpt_ref
    from ssl4gie_amd.Models.DPT_decoder import DPT_decoder
    m = DPT_decoder(num_classes=1, dense="depth")
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == dpt_ref.dpt_param_shapes()
    assert sum(p.numel() for p in m.parameters()) == 20065921


# ------------------------------------------------------------------ MoCo-v3 glue
def _g8_sd(g, tag):
    pre = f"{tag}/sd/"
    sd = {k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)}
    for k in g[f"{tag}/keys"]:
        k = str(k)
        if k.endswith("running_mean"):  # BatchNorm slots are recognised by their buffers
            sd[k] = torch.zeros(1)
    return sd


@pytest.mark.parametrize("tag", ["proj2", "pred2", "proj3"])
def test_g8_mlp_oracle_matches_reference(tag):
    from oracle import moco_ref
    g = load_golden("g8_moco.npz")
    sd = {k: (v.requires_grad_(True) if "running" not in k else v) for k, v in _g8_sd(g, tag).items()}
    x = torch.from_numpy(g[f"{tag}/x"]).requires_grad_(True)
    y = moco_ref.mlp_forward(sd, "", x)
    y.backward(torch.from_numpy(g[f"{tag}/dy"]))
    assert rel_err(y, g[f"{tag}/y"]) < 1e-5
    assert rel_err(x.grad, g[f"{tag}/dx"]) < 1e-4
    for k in g.files:
        if k.startswith(f"{tag}/grad/"):
            assert rel_err(sd[k[len(tag) + 6:]].grad, g[k]) < 1e-4, k


def test_g8_contrastive_loss_and_lars():
    from oracle import moco_ref
    from ssl4gie_amd.Models.moco_v3.moco.optimizer import LARS
    g = load_golden("g8_moco.npz")
    q = torch.from_numpy(g["cl/q"]).requires_grad_(True)
    loss = moco_ref.contrastive_loss(q, torch.from_numpy(g["cl/k"]), float(g["cl/T"]))
    loss.backward()
    assert abs(float(loss) - float(g["cl/loss"])) < 1e-6
    assert rel_err(q.grad, g["cl/dq"]) < 1e-5
    # LARS: oracle restatement and the engine's host-side optimizer, both against the reference's steps
    ps = [torch.from_numpy(g[f"lars/p0/{i}"]).clone() for i in range(3)]
    mus = [torch.zeros_like(p) for p in ps]
    host = [torch.nn.Parameter(p.clone()) for p in ps]
    opt = LARS(host, lr=0.3, weight_decay=1e-2, momentum=0.9)
    for step in range(3):
        grads = [torch.from_numpy(g[f"lars/g{step}/{i}"]) for i in range(3)]
        ps, mus = moco_ref.lars_step(ps, grads, mus, 0.3, 1e-2)
        for p, gr in zip(host, grads):
            p.grad = gr.clone()
        opt.step()
        for i in range(3):
            assert rel_err(ps[i], g[f"lars/p{step + 1}/{i}"]) < 1e-6 or float(ps[i].abs().max()) == 0
            assert torch.allclose(host[i].detach(), torch.from_numpy(g[f"lars/p{step + 1}/{i}"]), rtol=1e-5, atol=1e-7)


def test_moco_schema_and_trainable_count():
    from functools import partial
    from ssl4gie_amd.Models.moco_v3.moco import builder
    from ssl4gie_amd.Models.resnet import resnet50
    m = builder.MoCo_ResNet(partial(resnet50, zero_init_residual=True), 256, 4096, 1.0)
    assert sum(p.numel() for p in m.parameters() if p.requires_grad) == 35058752
    ks = set(m.state_dict())
    assert {"base_encoder.fc.0.weight", "base_encoder.fc.4.running_mean", "momentum_encoder.fc.3.weight",
            "predictor.0.weight", "predictor.3.weight"} <= ks
    assert not any(k.startswith("predictor.4") for k in ks)  # predictor has no last BN
    assert float(m.base_encoder.layer1[0].bn3.weight.abs().max()) == 0.0  # zero_init_residual
    assert all(not p.requires_grad for p in m.momentum_encoder.parameters())
