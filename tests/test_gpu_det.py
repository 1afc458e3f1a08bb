"""GPU parity of the detection ViT backbone: the pyramid's map kernels against
torch fp32 of the same op, ViTDet_FPN against the oracle restatement, and the whole det=True trunk
(window-ordered tokens, windowed + global blocks, streaming attention at N = 1024) against the
oracle that permutes / un-permutes in every windowed block like the reference does."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
F32, BF = torch.float32, torch.bfloat16


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ssl4gie_amd import _lib
    _lib.load()


def G(seed):
    return torch.Generator("cpu").manual_seed(seed)


@pytest.mark.parametrize("dt,tol", [(F32, 1e-6), (BF, 1e-2)])
def test_maxpool2x2_and_gelu_map(dt, tol):
    from ssl4gie_amd import ops
    x = torch.randn(2, 16, 12, 10, generator=G(1)).to(dt)  # NCHW
    xd = x.permute(0, 2, 3, 1).contiguous().to(DEV)
    xr = x.float().clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 2)
    y = ops.maxpool2x2_fwd(xd)
    assert torch.equal(y.float().cpu().permute(0, 3, 1, 2), yr.detach())
    dy = torch.randn(yr.shape, generator=G(2)).to(dt)
    yr.backward(dy.float())
    dx = ops.maxpool2x2_bwd(xd, dy.permute(0, 2, 3, 1).contiguous().to(DEV))
    assert torch.equal(dx.float().cpu().permute(0, 3, 1, 2), xr.grad)
    xr2 = x.float().clone().requires_grad_(True)
    gr = F.gelu(xr2)
    g = ops.gelu_map(xd)
    assert rel_err(g.float().cpu().permute(0, 3, 1, 2), gr.detach()) < max(tol, 4e-3 if dt == BF else 0)
    gr.backward(dy.new_ones(gr.shape).float())
    dg = ops.gelu_map(xd, torch.ones_like(xd))
    assert rel_err(dg.float().cpu().permute(0, 3, 1, 2), xr2.grad) < max(tol, 4e-3 if dt == BF else 0)


@pytest.mark.parametrize("dt,tol", [(F32, 2e-5), (BF, 1e-2)])
def test_map_layernorm_fwd_bwd(dt, tol):
    from ssl4gie_amd import ops
    B, C, H, W = 3, 24, 10, 12
    x = (torch.randn(B, C, H, W, generator=G(3)) * 2 + 5).to(dt)  # |mean| >> 0: exercises the pivot
    w = 1 + 0.2 * torch.randn(C, H, W, generator=G(4))
    b = 0.3 * torch.randn(C, H, W, generator=G(5))
    dy = torch.randn(B, C, H, W, generator=G(6)).to(dt)
    xr = x.float().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.layer_norm(xr, (C, H, W), wr, br, 1e-5)
    yr.backward(dy.float())
    hwc = lambda t: t.permute(1, 2, 0).contiguous().to(DEV)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(DEV)
    y, mean, rstd = ops.map_layernorm_fwd(nhwc(x), hwc(w), hwc(b), 1e-5)
    assert rel_err(y.float().cpu().permute(0, 3, 1, 2), yr.detach()) < tol
    assert rel_err(mean.cpu(), x.float().mean((1, 2, 3))) < 1e-5
    dw = torch.empty(H, W, C, device=DEV)
    db = torch.empty(H, W, C, device=DEV)
    dx = ops.map_layernorm_bwd(nhwc(x), nhwc(dy), hwc(w), mean, rstd, dw, db)
    assert rel_err(dx.float().cpu().permute(0, 3, 1, 2), xr.grad) < tol * 3
    assert rel_err(dw.cpu().permute(2, 0, 1), wr.grad) < tol
    assert rel_err(db.cpu().permute(2, 0, 1), br.grad) < tol


def test_window_permutation_is_the_reference_construction():
    """integer work, exact: perm / inv_perm of WindowedAttention for N = 4096 (and 1024), window 16"""
    from oracle import det_ref
    from ssl4gie_amd.Models.models import window_permutation
    for s in (64, 32, 16):
        perm, inv, windows = det_ref.window_perm(s * s, 16)
        p2, i2 = window_permutation(s, 16)
        assert torch.equal(perm, p2) and torch.equal(inv, i2) and windows == (s // 16) ** 2
        assert torch.equal(perm[inv], torch.arange(s * s))
        # every run of 256 indices is one 16 x 16 window of the grid
        w0 = perm[:256].reshape(16, 16)
        assert torch.equal(w0, torch.arange(16)[:, None] * s + torch.arange(16)[None, :])


def _fpn_state(m, seed):
    g = G(seed)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if p.dim() == 4:
                fan = p.shape[1] * p.shape[2] * p.shape[3] if "fpn3.0" not in name and "fpn4.0" not in name \
                    and "fpn4.3" not in name else p.shape[0]
                p.copy_(torch.randn(p.shape, generator=g) / fan ** 0.5)
            elif p.dim() == 3:  # LayerNorm((C, H, W)) affine
                p.copy_((1 + 0.1 * torch.randn(p.shape, generator=g)) if name.endswith("weight")
                        else 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.1 * torch.randn(p.shape, generator=g))


@pytest.mark.parametrize("prec,tol,gtol,arena", [("fp32", 1e-3, 5e-3, False), ("bf16", 4e-2, 1.5e-1, False),
                                                 ("fp32", 1e-3, 5e-3, True)])
def test_vitdet_fpn_vs_oracle(prec, tol, gtol, arena):
    """arena: parameters and gradients re-homed into ParamArena slices (ArenaAdamW) — the (C, H, W) LayerNorm
    tables keep their channels-last memory there and the kernels read / write them in place"""
    from oracle import det_ref
    from ssl4gie_amd.Models.models import ViTDet_FPN
    torch.manual_seed(0)
    m = ViTDet_FPN(grid=16, dim=768)
    _fpn_state(m, 7)
    sd = {"fpn." + k: v.detach().clone().contiguous().requires_grad_(True) for k, v in m.state_dict().items()}
    m.to(DEV).set_precision(prec)
    if arena:
        from ssl4gie_amd.optim import ArenaAdamW
        opt = ArenaAdamW(m, list(m.parameters()), lr=1e-4)
        w = m.fpn1[2].weight
        assert m.arena().owns(w) and not w.is_contiguous() and w.permute(1, 2, 0).is_contiguous()
    x = torch.randn(2, 256, 768, generator=G(8))
    xd = x.to(DEV).requires_grad_(True)
    out = m(xd)
    xo = x.clone().requires_grad_(True)
    ref = det_ref.fpn(sd, xo)
    assert list(out.keys()) == ["0", "1", "2", "3", "pool"]
    loss, loss_o = 0, 0
    for i, k in enumerate(out):
        assert out[k].shape == ref[k].shape and out[k].dtype == torch.float32
        assert rel_err(out[k].cpu(), ref[k].detach()) < tol, k
        wgt = torch.randn(ref[k].shape, generator=G(20 + i))
        loss = loss + (out[k] * wgt.to(DEV)).sum()
        loss_o = loss_o + (ref[k] * wgt).sum()
    loss.backward()
    loss_o.backward()
    assert rel_err(xd.grad.cpu(), xo.grad) < gtol
    pg = dict(m.named_parameters())
    for name in ("fpn1.1.weight", "fpn1.2.weight", "fpn2.3.bias", "fpn3.0.weight", "fpn4.1.weight",
                 "fpn4.3.bias", "fpn4.5.weight", "fpn4.6.weight", "fpn4.7.bias"):
        assert rel_err(pg[name].grad.cpu(), sd["fpn." + name].grad) < gtol, name
    if arena:
        g = m.fpn4[5].weight.grad
        assert g.data_ptr() == m.arena().grad_view(m.fpn4[5].weight).data_ptr() and g.permute(1, 2, 0).is_contiguous()


@pytest.mark.parametrize("prec,tol", [("fp32", 2e-3), ("bf16", 6e-2)])
def test_det_trunk_windowed_and_global_blocks_vs_oracle(prec, tol):
    """VisionTransformer_from_Any(det=True, fixed_size=512): N = 1024 tokens, 4 windows; output tokens
    (after the final norm, back in row-major order) and gradients vs the oracle"""
    from oracle import det_ref
    from ssl4gie_amd.Models import models
    torch.manual_seed(0)
    m = models.VisionTransformer_from_Any(False, 0, False, None, True, 512, 768, 12, 12, "cls")
    assert "cls_token" not in m.state_dict() and m.out_channels == 256
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in m.state_dict().items()}
    m.to(DEV).set_precision(prec)
    imgs = torch.randn(1, 3, 512, 512, generator=G(9))
    tok = m.forward_features(imgs.to(DEV))
    ref = det_ref.det_trunk(sd, imgs, 512)
    assert tok.shape == (1, 1024, 768)
    assert rel_err(tok.cpu(), ref.detach()) < tol
    wgt = torch.randn(ref.shape, generator=G(10))
    (tok * wgt.to(DEV)).sum().backward()
    (ref * wgt).sum().backward()
    pg = dict(m.named_parameters())
    gt = 5e-3 if prec == "fp32" else 0.25
    for name in ("pos_embed", "patch_embed.proj.weight", "blocks.0.attn.qkv.weight", "blocks.2.attn.qkv.weight",
                 "blocks.10.attn.proj.weight", "blocks.11.mlp.fc1.weight", "norm.weight"):
        assert rel_err(pg[name].grad.cpu(), sd[name].grad) < gt, name
    out = m(imgs.to(DEV))  # backbone + pyramid, the dict torchvision's FasterRCNN consumes
    assert [tuple(v.shape) for v in out.values()] == [(1, 256, 128, 128), (1, 256, 64, 64), (1, 256, 32, 32),
                                                       (1, 256, 16, 16), (1, 256, 8, 8)]
