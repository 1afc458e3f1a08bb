"""GPU parity of the fused loss kernels (csrc/loss_ops.hip) against the reference's
own ScaleAndShiftInvariantLoss (fixture G7: value + gradient from Depth_estimation/Metrics/losses.py,
incl. an image without valid pixels) and against the host-side torch formulations (themselves pinned
by G7 / G9) on random maps with masked regions, odd sizes and every scale count."""
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


def test_fused_ssi_loss_matches_reference_fixture():
    from ssl4gie_amd.losses import ScaleAndShiftInvariantLoss
    g = load_golden("g7_ssi_loss.npz")
    pred = torch.from_numpy(g["pred"]).to(DEV).requires_grad_(True)
    target = torch.from_numpy(g["target"]).to(DEV)
    loss = ScaleAndShiftInvariantLoss(alpha=0.1)(pred, target)
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    assert rel_err(pred.grad, g["grad"]) < 1e-4


@pytest.mark.parametrize("B,H,W,alpha,scales", [(4, 224, 224, 0.1, 4), (3, 50, 37, 0.5, 4), (2, 64, 64, 0.1, 2),
                                                  (2, 33, 40, 0.0, 4), (5, 16, 16, 1.0, 1)])
def test_fused_ssi_loss_matches_host_formulation(B, H, W, alpha, scales, monkeypatch):
    from ssl4gie_amd.losses import ScaleAndShiftInvariantLoss
    gen = torch.Generator("cpu").manual_seed(B * 1000 + H)
    pred = torch.rand(B, 1, H, W, generator=gen)
    target = torch.rand(B, 1, H, W, generator=gen)
    target = torch.where(torch.rand(B, 1, H, W, generator=gen) < 0.25, torch.zeros(()), target)
    target[0, :, : H // 3] = 0            # a masked band: neighbour pairs across the mask edge
    if B > 2:
        target[2] = 0                     # no valid pixel at all
    res = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("SSL4GIE_FUSED_LOSS", fused)
        p = pred.double().to(DEV).requires_grad_(True) if fused == "0" else pred.to(DEV).requires_grad_(True)
        fn = ScaleAndShiftInvariantLoss(alpha=alpha, scales=scales)
        if fused == "0":  # fp64 torch formulation as the yardstick
            import ssl4gie_amd.losses as Lm
            pr, tg = p.squeeze(1), target.double().to(DEV).squeeze(1)
            mask = tg > 0
            s, h = Lm.compute_scale_and_shift(pr, tg, mask)
            ssi = s.view(-1, 1, 1) * pr + h.view(-1, 1, 1)
            loss = Lm.mse_loss(ssi, tg, mask)
            for k in range(scales if alpha > 0 else 0):
                st = 2 ** k
                loss = loss + alpha * Lm.gradient_loss(ssi[:, ::st, ::st], tg[:, ::st, ::st], mask[:, ::st, ::st])
        else:
            loss = fn(p, target.to(DEV))
        (loss * 3.0).backward()
        res[fused] = (float(loss.detach()), p.grad.detach().double().cpu())
    assert abs(res["1"][0] - res["0"][0]) < 2e-5 * max(abs(res["0"][0]), 1e-6)
    assert rel_err(res["1"][1], res["0"][1]) < 2e-4


@pytest.mark.parametrize("B,n", [(4, 224 * 224), (3, 1000), (1, 77)])
def test_fused_dice_loss_matches_host_formulation(B, n, monkeypatch):
    from ssl4gie_amd.losses import SoftDiceLoss
    gen = torch.Generator("cpu").manual_seed(n)
    logits = torch.randn(B, 1, n, generator=gen) * 2
    target = (torch.rand(B, 1, n, generator=gen) < 0.3).float()
    res = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("SSL4GIE_FUSED_LOSS", fused)
        l = (logits.double() if fused == "0" else logits).to(DEV).requires_grad_(True)
        loss = SoftDiceLoss()(l, (target.double() if fused == "0" else target).to(DEV))
        loss.backward()
        res[fused] = (float(loss.detach()), l.grad.detach().double().cpu())
    assert abs(res["1"][0] - res["0"][0]) < 1e-5
    assert rel_err(res["1"][1], res["0"][1]) < 1e-4
