"""GPU parity of the whole MAE / ViT path against golden vectors produced by the REFERENCE classes
(tests/golden/make_golden.py) and against the CPU oracle.  fp32 engine: <= 1e-3 rel (north_star);
mask / ids bit-exact; bf16 engine: loss-level agreement."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ssl4gie_amd import _lib
    _lib.load()


def _build(cfg, seed, precision):
    from functools import partial
    import torch.nn as nn
    from oracle import synth
    from ssl4gie_amd.Models.mae.models_mae import MaskedAutoencoderViT
    m = MaskedAutoencoderViT(img_size=cfg.img_size, patch_size=cfg.patch_size,
                             in_chans=cfg.in_chans, embed_dim=cfg.embed_dim, depth=cfg.depth,
                             num_heads=cfg.num_heads, decoder_embed_dim=cfg.decoder_embed_dim,
                             decoder_depth=cfg.decoder_depth,
                             decoder_num_heads=cfg.decoder_num_heads, mlp_ratio=cfg.mlp_ratio,
                             norm_layer=partial(nn.LayerNorm, eps=cfg.ln_eps),
                             norm_pix_loss=cfg.norm_pix_loss)
    sd = synth.mae_state_dict(cfg, seed)
    m.load_state_dict(sd, strict=True)
    m.to(DEV).set_precision(precision)
    return m, sd


@pytest.mark.parametrize("tag,npl", [("raw", False), ("npl", True)])
def test_mae_tiny_fp32_matches_reference_golden(tag, npl):
    from oracle import mae_ref, synth
    g = load_golden("g5_mae_tiny.npz")
    cfg = mae_ref.MAEConfig(**{**mae_ref.TINY.__dict__, "norm_pix_loss": npl})
    m, sd = _build(cfg, 1, "fp32")
    assert synth.state_dict_digest(sd) == str(g[f"{tag}/digest"])
    imgs = synth.synth_images(4, cfg, seed=1).to(DEV)
    noise = torch.from_numpy(synth.synth_noise(4, cfg.num_patches, seed=1)).to(DEV)
    loss, pred, mask = m(imgs, mask_ratio=0.75, noise=noise)
    loss.backward()
    assert np.array_equal(mask.cpu().numpy(), g[f"{tag}/mask"])          # bit-exact
    ref_loss = float(g[f"{tag}/loss"])
    assert abs(float(loss) - ref_loss) < 1e-4 * abs(ref_loss)
    assert rel_err(pred, g[f"{tag}/pred"]) < 1e-3
    n = 0
    for name, p in m.named_parameters():
        key = f"{tag}/grad/{name}"
        if key in g.files:
            assert p.grad is not None, name
            assert rel_err(p.grad, g[key]) < 1e-3, name
            n += 1
    assert n >= 60
    # arena: gradients alias the flat buffer (what the DDP buckets rely on)
    a = m.arena()
    assert all(p.grad.data_ptr() == a.grad_view(p).data_ptr()
               for p in m.parameters() if p.grad is not None)


def test_mae_vitb_fp32_matches_reference_golden():
    from oracle import mae_ref, synth
    g = load_golden("g5_mae_vitb.npz")
    cfg = mae_ref.MAEConfig(**{**mae_ref.VIT_B.__dict__, "norm_pix_loss": True})
    m, sd = _build(cfg, 0, "fp32")
    assert synth.state_dict_digest(sd) == str(g["digest"])
    imgs = synth.synth_images(2, cfg, seed=0).to(DEV)
    noise = torch.from_numpy(synth.synth_noise(2, 196, seed=0)).to(DEV)
    loss, pred, mask = m(imgs, noise=noise)
    loss.backward()
    assert np.array_equal(mask.cpu().numpy(), g["mask"])
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * float(g["loss"])
    assert rel_err(pred, g["pred"]) < 1e-3
    grads = {n: p.grad for n, p in m.named_parameters() if p.grad is not None}
    names = [str(s) for s in g["grad_names"]]
    assert sorted(grads) == sorted(names)
    norms = dict(zip(names, g["grad_norms"]))
    for n_, gr in grads.items():
        assert abs(float(gr.norm()) - norms[n_]) < 1e-3 * max(norms[n_], 1e-6), n_
        if f"grad/{n_}" in g.files:
            assert rel_err(gr, g[f"grad/{n_}"]) < 1e-3, n_
        else:
            sl = gr.reshape(gr.shape[0], -1)[:8, :64]
            ref = g[f"gslice/{n_}"]
            assert (sl.cpu() - torch.from_numpy(ref)).abs().max() < 1e-3 * float(gr.abs().max()), n_


def test_mae_vitb_bf16_close_to_reference():
    from oracle import mae_ref, synth
    g = load_golden("g5_mae_vitb.npz")
    cfg = mae_ref.MAEConfig(**{**mae_ref.VIT_B.__dict__, "norm_pix_loss": True})
    m, _ = _build(cfg, 0, "bf16")
    imgs = synth.synth_images(2, cfg, seed=0).to(DEV)
    noise = torch.from_numpy(synth.synth_noise(2, 196, seed=0)).to(DEV)
    loss, pred, mask = m(imgs, noise=noise)
    loss.backward()
    assert np.array_equal(mask.cpu().numpy(), g["mask"])
    assert abs(float(loss) - float(g["loss"])) < 5e-3 * float(g["loss"])
    assert rel_err(pred, g["pred"]) < 5e-2
    norms = dict(zip([str(s) for s in g["grad_names"]], g["grad_norms"]))
    bad = [n for n, p in m.named_parameters()
           if p.grad is not None and abs(float(p.grad.norm()) - norms[n]) > 0.1 * max(norms[n], 1e-6)]
    assert not bad, bad[:8]


def _train_curve(cfg, precision, steps, g, optim="torch"):
    """optim: "torch" = torch.optim.AdamW (the reference driver's, main_pretrain.py:179-180);
    "arena" = ssl4gie_amd.optim.ArenaAdamW, bench.py's default, incl. the bf16 operand copies it writes"""
    from oracle import synth
    m, _ = _build(cfg, 0, precision)
    decay = [p for n, p in m.named_parameters() if p.requires_grad and p.ndim > 1 and not n.endswith(".bias")]
    no_decay = [p for n, p in m.named_parameters() if p.requires_grad and not (p.ndim > 1 and not n.endswith(".bias"))]
    groups = [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": 0.05}]
    if optim == "arena":
        from ssl4gie_amd.optim import ArenaAdamW
        opt = ArenaAdamW(m, groups, lr=float(g["lr"]), betas=(0.9, 0.95))
    else:
        opt = torch.optim.AdamW(groups, lr=float(g["lr"]), betas=(0.9, 0.95))
    b = int(g["batch"])
    losses = []
    for it in range(steps):
        imgs = synth.synth_images(b, cfg, seed=it % 4).to(DEV)
        noise = torch.from_numpy(synth.synth_noise(b, cfg.num_patches, seed=100 + it)).to(DEV)
        opt.zero_grad(set_to_none=True)
        loss, _, _ = m(imgs, noise=noise)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    return np.array(losses)


def test_loss_curve_tiny_fp32_100_steps():
    """north_star: losses match the CPU reference within 1e-3 over 100 steps."""
    from oracle import mae_ref
    g = load_golden("g5_curve_tiny.npz")
    cfg = mae_ref.MAEConfig(**{**mae_ref.TINY.__dict__, "norm_pix_loss": True})
    losses = _train_curve(cfg, "fp32", 100, g)
    err = np.abs(losses - g["losses"]) / g["losses"]
    assert err.max() < 1e-3, (int(err.argmax()), float(err.max()))


def test_loss_curve_vitb_fp32_100_steps():
    from oracle import mae_ref
    g = load_golden("g5_curve_vitb.npz")
    cfg = mae_ref.MAEConfig(**{**mae_ref.VIT_B.__dict__, "norm_pix_loss": True})
    losses = _train_curve(cfg, "fp32", 100, g)
    err = np.abs(losses - g["losses"]) / g["losses"]
    assert err.max() < 1e-3, (int(err.argmax()), float(err.max()))


def test_loss_curve_vitb_fp32_arena_adamw_100_steps():
    """the optimizer bench.py actually runs (ArenaAdamW: one kernel over the arena, also emitting the bf16
    operand copies) against the reference's 100-step curve, at the north_star's 1e-3"""
    from oracle import mae_ref
    g = load_golden("g5_curve_vitb.npz")
    cfg = mae_ref.MAEConfig(**{**mae_ref.VIT_B.__dict__, "norm_pix_loss": True})
    losses = _train_curve(cfg, "fp32", 100, g, optim="arena")
    err = np.abs(losses - g["losses"]) / g["losses"]
    assert err.max() < 1e-3, (int(err.argmax()), float(err.max()))


# bf16 production engine over the SAME 100 steps as the fp32 gate.  Measured on MI355X (round 3,
# profiles/r03e_loss_curves.log): max |loss - reference| / reference over the 100 steps = 2.17e-4 with
# torch.optim.AdamW and 2.18e-4 with ArenaAdamW (at step 1; mean 1.8e-5, final step 2e-6) — inside the
# north_star's 1e-3 even at bf16 operand precision.  The bar is 1.5x the measurement.
BF16_CURVE_MEASURED = 2.2e-4
BF16_CURVE_BAR = 1.5 * BF16_CURVE_MEASURED


@pytest.mark.parametrize("optim", ["torch", "arena"])
def test_loss_curve_vitb_bf16_100_steps(optim):
    from oracle import mae_ref
    g = load_golden("g5_curve_vitb.npz")
    cfg = mae_ref.MAEConfig(**{**mae_ref.VIT_B.__dict__, "norm_pix_loss": True})
    losses = _train_curve(cfg, "bf16", 100, g, optim=optim)
    err = np.abs(losses - g["losses"]) / g["losses"]
    print(f"bf16 curve [{optim}]: max rel deviation {err.max():.3e} at step {int(err.argmax())}, "
          f"mean {err.mean():.3e}, final {err[-1]:.3e}")
    assert err.max() < BF16_CURVE_BAR, (int(err.argmax()), float(err.max()))
    assert losses[-1] < 0.75 * losses[0]


def test_vit_trunk_taps_and_head_fp32():
    """ViT_from_MAE trunk (reference models.py:427-475): taps after blocks 2/5/8/11, cls readout +
    linear head, against the oracle with the same weights."""
    from oracle import mae_ref, synth
    from ssl4gie_amd import utils
    cfg = mae_ref.VIT_B
    m = utils.get_MAE_backbone(None, True, 6, False, None, False)
    sd = synth.mae_state_dict(cfg, 3)
    m.load_my_state_dict(sd)
    m.to(DEV).set_precision("fp32")
    imgs = synth.synth_images(2, cfg, seed=5)
    with torch.no_grad():
        taps = m.forward_features(imgs.to(DEV), dense="depth")
        ref_taps = mae_ref.vit_trunk(sd, cfg, imgs, dense=True)
    assert len(taps) == 4
    for t, r in zip(taps, ref_taps):
        assert rel_err(t, r) < 1e-3
    out = m(imgs.to(DEV))
    lat = mae_ref.vit_trunk(sd, cfg, imgs, dense=False)[:, 0]
    ref = torch.nn.functional.linear(lat, m.lin_head.weight.detach().cpu(), m.lin_head.bias.detach().cpu())
    assert out.shape == (2, 6) and rel_err(out.detach(), ref.detach()) < 1e-3
    out.sum().backward()
    assert m.lin_head.weight.grad is not None and m.blocks[0].attn.qkv.weight.grad is not None
    assert m.pos_embed.grad is None


def test_patchify_api_exact():
    from ssl4gie_amd.Models.mae import models_mae
    g = load_golden("g2_patchify.npz")
    m = models_mae.MaskedAutoencoderViT(img_size=64, embed_dim=64, depth=1, num_heads=1,
                                        decoder_embed_dim=64, decoder_depth=1, decoder_num_heads=2)
    p = m.patchify(torch.from_numpy(g["imgs"]).to(DEV))
    assert np.array_equal(p.cpu().numpy(), g["patches"])
    assert torch.equal(m.unpatchify(p).cpu(), torch.from_numpy(g["imgs"]))
