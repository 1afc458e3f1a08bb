"""GPU parity of the DPT depth decoder path: channels-last glue kernels
against torch fp32 of the same op, the decoder against the golden fixture generated from the
REFERENCE's DPT_decoder class (fp32 engine <= 1e-3 rel), and ViT_from_MAE(dense="depth") end to end
against the CPU oracle."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import ROOT, load_golden, rel_err

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


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("dt", [F32, BF])
@pytest.mark.parametrize("stride,relu", [(1, False), (1, True), (2, False)])
def test_im2col3x3_equals_unfold(dt, stride, relu):
    from ssl4gie_amd import ops
    x = torch.randn(2, 16, 9, 11, generator=G(1)).to(dt)  # NCHW
    cols = ops.im2col3x3(nhwc(x).to(DEV), stride, relu).float().cpu()
    xx = F.relu(x.float()) if relu else x.float()
    ref = F.unfold(xx, 3, padding=1, stride=stride)  # [B, C*9, L] with (c, dy, dx) order
    B, _, L = ref.shape
    ref = ref.view(B, 16, 9, L).permute(0, 3, 2, 1).reshape(B * L, 9 * 16)  # (dy*3+dx, c)
    assert torch.equal(cols[:, :144], ref)
    assert torch.count_nonzero(cols[:, 144:]) == 0


@pytest.mark.parametrize("shape", [(2, 16, 7, 5), (3, 256, 14, 14), (2, 128, 9, 33), (1, 8, 1, 1)])
@pytest.mark.parametrize("dt,tol", [(F32, 3e-6), (BF, 1e-2)])  # fp32: the source coordinate itself is rounded
def test_bilinear2x_fwd_bwd(dt, tol, shape):
    from ssl4gie_amd import ops
    x = torch.randn(*shape, generator=G(2)).to(dt)
    y = ops.bilinear2x_fwd(nhwc(x).to(DEV))
    xr = x.float().requires_grad_(True)
    yr = F.interpolate(xr, scale_factor=2, mode="bilinear", align_corners=True)
    assert rel_err(nchw(y.float().cpu()), yr) < tol
    dy = torch.randn(yr.shape, generator=G(3)).to(dt)
    yr.backward(dy.float())
    dx = ops.bilinear2x_bwd(nhwc(dy).to(DEV))
    assert rel_err(nchw(dx.float().cpu()), xr.grad) < tol


def _conv_case(dt, Cin, Cout, H, W, stride, relu, bias):
    from ssl4gie_amd.dpt_engine import Conv3x3Fn
    from ssl4gie_amd.engine import GradSink, LPCache
    x = torch.randn(2, Cin, H, W, generator=G(4))
    w = torch.randn(Cout, Cin, 3, 3, generator=G(5)) / (9 * Cin) ** 0.5
    b = 0.1 * torch.randn(Cout, generator=G(6)) if bias else None
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    yr = F.conv2d(F.relu(xr) if relu else xr, wr, br, stride=stride, padding=1)
    dy = torch.randn(yr.shape, generator=G(7))
    yr.backward(dy)
    xd = nhwc(x).to(DEV).to(dt).requires_grad_(True)
    wd = torch.nn.Parameter(w.to(DEV))
    bd = torch.nn.Parameter(b.to(DEV)) if bias else None
    y = Conv3x3Fn.apply(xd, wd, bd, stride, relu, GradSink(None), LPCache())
    y.backward(nhwc(dy).to(DEV).to(dt))
    return (nchw(y.detach().float().cpu()), yr.detach(), nchw(xd.grad.float().cpu()), xr.grad,
            wd.grad.cpu(), wr.grad, None if not bias else bd.grad.cpu(), None if not bias else br.grad)


@pytest.mark.parametrize("Cin,Cout,H,W,stride,relu,bias",
                         [(16, 32, 9, 7, 1, False, False), (32, 16, 8, 8, 1, True, True),
                          (96, 64, 12, 12, 1, False, False), (24, 24, 14, 14, 2, False, True)])
def test_conv3x3_fp32_fwd_bwd(Cin, Cout, H, W, stride, relu, bias):
    y, yr, dx, dxr, dw, dwr, db, dbr = _conv_case(F32, Cin, Cout, H, W, stride, relu, bias)
    assert rel_err(y, yr) < 1e-5
    assert rel_err(dx, dxr) < 1e-5
    assert rel_err(dw, dwr) < 1e-5
    if bias:
        assert rel_err(db, dbr) < 1e-5


def test_conv3x3_bf16_fast_paths():
    """large enough for the 256x256 NT / TN kernels (K padded to whole 64-deep tiles)"""
    y, yr, dx, dxr, dw, dwr, db, dbr = _conv_case(BF, 96, 256, 56, 56, 1, True, True)
    assert rel_err(y, yr) < 1.5e-2
    assert rel_err(dx, dxr) < 1.5e-2
    assert rel_err(dw, dwr) < 1.5e-2
    assert rel_err(db, dbr) < 1.5e-2


@pytest.mark.parametrize("k", [2, 4])
def test_conv_transpose_fp32(k):
    from ssl4gie_amd.dpt_engine import ConvTransposeFn
    from ssl4gie_amd.engine import GradSink, LPCache
    B, H, W, Cin, Cout = 2, 5, 6, 24, 16
    x = torch.randn(B, Cin, H, W, generator=G(8))
    w = torch.randn(Cin, Cout, k, k, generator=G(9)) / Cin ** 0.5
    b = 0.1 * torch.randn(Cout, generator=G(10))
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = F.conv_transpose2d(xr, wr, br, stride=k)
    dy = torch.randn(yr.shape, generator=G(11))
    yr.backward(dy)
    x2 = nhwc(x).reshape(-1, Cin).to(DEV).requires_grad_(True)
    wd, bd = torch.nn.Parameter(w.to(DEV)), torch.nn.Parameter(b.to(DEV))
    y = ConvTransposeFn.apply(x2, wd, bd, B, H, W, GradSink(None), LPCache())
    y.backward(nhwc(dy).to(DEV))
    assert rel_err(nchw(y.detach().cpu()), yr) < 1e-5
    assert rel_err(x2.grad.cpu().view(B, H, W, Cin).permute(0, 3, 1, 2), xr.grad) < 1e-5
    assert rel_err(wd.grad.cpu(), wr.grad) < 1e-5
    assert rel_err(bd.grad.cpu(), br.grad) < 1e-5


def test_depth_head_fwd_bwd():
    from ssl4gie_amd.dpt_engine import DepthHeadFn
    from ssl4gie_amd.engine import GradSink
    x = torch.randn(2, 32, 10, 12, generator=G(12))
    w = torch.randn(1, 32, 1, 1, generator=G(13)) * 0.3
    b = torch.randn(1, generator=G(14))
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = torch.sigmoid(F.conv2d(F.relu(xr), wr, br))
    dy = torch.randn(yr.shape, generator=G(15))
    yr.backward(dy)
    xd = nhwc(x).to(DEV).requires_grad_(True)
    wd, bd = torch.nn.Parameter(w.to(DEV)), torch.nn.Parameter(b.to(DEV))
    y = DepthHeadFn.apply(xd, wd, bd, GradSink(None))
    y.backward(dy.to(DEV))
    assert rel_err(y.detach().cpu(), yr) < 1e-6
    assert rel_err(nchw(xd.grad.cpu()), xr.grad) < 1e-5
    assert rel_err(wd.grad.cpu(), wr.grad) < 1e-5
    assert rel_err(bd.grad.cpu(), br.grad) < 1e-5


def _dpt_inputs(seed, b=2):
    g = torch.Generator("cpu").manual_seed(seed)
    acts = [torch.randn(b, 197, 768, generator=g) for _ in range(4)]
    target = torch.rand(b, 1, 224, 224, generator=g)
    target = torch.where(torch.rand(b, 1, 224, 224, generator=g) < 0.1, torch.zeros(()), target)
    return acts, target


@pytest.mark.parametrize("prec,tol,gtol", [("fp32", 1e-3, 2e-3), ("bf16", 3e-2, 1.5e-1)])
def test_dpt_decoder_matches_reference_golden(prec, tol, gtol):
    """engine DPT_decoder vs outputs / gradients of the reference's own class (g6 fixture)"""
    from oracle import dpt_ref
    from ssl4gie_amd.Models.DPT_decoder import DPT_decoder
    from ssl4gie_amd.losses import ScaleAndShiftInvariantLoss
    g = load_golden("g6_dpt_depth.npz")
    m = DPT_decoder(num_classes=1, dense="depth")
    m.load_state_dict(dpt_ref.dpt_state_dict(int(g["seed_weights"])), strict=True)
    m.to(DEV).set_precision(prec)
    acts, target = _dpt_inputs(int(g["seed_inputs"]))
    acts = [a.to(DEV).requires_grad_(True) for a in acts]
    out = m(acts)
    loss = ScaleAndShiftInvariantLoss(alpha=0.1)(out, target.to(DEV))
    loss.backward()
    assert out.shape == (2, 1, 224, 224)
    assert rel_err(out, g["out"]) < tol
    assert abs(float(loss.detach()) - float(g["loss"])) < tol * abs(float(g["loss"]))
    no_grad = set(str(n) for n in g["no_grad_params"])
    for name, p in m.named_parameters():
        if name in no_grad:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
    for name, ref_norm in zip(g["grad_names"], g["grad_norms"]):
        p = dict(m.named_parameters())[str(name)]
        got = float(p.grad.norm())
        assert abs(got - ref_norm) <= gtol * max(ref_norm, 1e-12), (str(name), got, ref_norm)
        key = f"grad/{name}"
        if prec == "fp32" and key in g.files:
            assert rel_err(p.grad, g[key]) < 2e-3, str(name)
    for i in range(4):
        # bf16: the SSI loss's L1 gradient-matching term makes dL/dpred ill-conditioned under ANY half-precision
        # rounding of the prediction — the reference's own bf16-autocast gradients are 0.19 (median) to 0.22 (worst
        # tensor) off its fp64 gradients on this model (G17, tests/golden/make_golden.py g17_bf16_bars): 1.5 x that
        assert rel_err(acts[i].grad[:, :4, :64], g[f"act_grad_slice/{i}"]) < (2e-3 if prec == "fp32" else 0.33)
        assert torch.count_nonzero(acts[i].grad[:, 0]) == 0  # the cls row is sliced away


@pytest.mark.parametrize("prec,tol,gtol", [("fp32", 1e-3, 5e-3), ("bf16", 5e-2, 2e-1)])
def test_dpt_seg_decoder_matches_reference_golden(prec, tol, gtol):
    """engine DPT_decoder(dense="seg") (BatchNorm fusion blocks, seg head) vs outputs / gradients of
    the reference's own class + SoftDiceLoss (g9 fixture; Dropout.p = 0 on both sides)"""
    from oracle import dpt_ref
    from ssl4gie_amd.Models.DPT_decoder import DPT_decoder
    from ssl4gie_amd.losses import SoftDiceLoss
    g = load_golden("g9_dpt_seg.npz")
    m = DPT_decoder(num_classes=1, dense="seg")
    m.load_state_dict(dpt_ref.seg_state_dict(int(g["seed_weights"])), strict=False)
    m.output_conv[3].p = 0.0
    m.to(DEV).set_precision(prec).train()
    acts, _ = _dpt_inputs(int(g["seed_inputs"]))
    acts = [a.to(DEV).requires_grad_(True) for a in acts]
    target = torch.from_numpy(g["target"]).float().to(DEV)
    out = m(acts)
    loss = SoftDiceLoss()(out, target)
    loss.backward()
    assert out.shape == (2, 1, 224, 224) and out.dtype == torch.float32
    assert rel_err(out, g["out"]) < tol
    assert abs(float(loss.detach()) - float(g["loss"])) < tol * abs(float(g["loss"]))
    params = dict(m.named_parameters())
    for name in g["no_grad_params"]:
        p = params[str(name)]
        assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
    for name, ref_norm in zip(g["grad_names"], g["grad_norms"]):
        got = float(params[str(name)].grad.norm())
        assert abs(got - ref_norm) <= gtol * max(ref_norm, 1e-12), (str(name), got, ref_norm)
        key = f"grad/{name}"
        if prec == "fp32" and key in g.files:
            # BatchNorm biases upstream of another BatchNorm receive a gradient that is a sum of
            # cancelling terms (~1e-5): fp32 summation order shows at the 1e-2 level there
            assert rel_err(params[str(name)].grad, g[key]) < 2e-2, str(name)
    for i in range(4):
        assert rel_err(acts[i].grad[:, :4, :64], g[f"act_grad_slice/{i}"]) < (5e-3 if prec == "fp32" else 0.25)
    if prec == "fp32":  # running statistics follow nn.BatchNorm2d's momentum update
        assert rel_err(m.output_conv[1].running_mean, g["running_mean/output_conv.1"]) < 1e-3


def test_vit_from_mae_depth_end_to_end_vs_oracle():
    """ViT_from_MAE(dense="depth") (tiny trunk config is not possible: DPT is fixed to 768 x 14 x 14)
    on a 2-block ViT-B-width trunk: forward + SSI loss + gradients vs the CPU oracle."""
    from oracle import dpt_ref, mae_ref, synth
    from ssl4gie_amd.Models import models
    from ssl4gie_amd.losses import ScaleAndShiftInvariantLoss
    torch.manual_seed(0)
    m = models.ViT_from_MAE(None, False, 1, False, "depth", False, None, 768, 12, 12, "cls")
    # seeded, O(1) decoder weights (default conv init makes the sigmoid saturate less informative)
    dsd = dpt_ref.dpt_state_dict(4)
    m.decoder.load_state_dict(dsd, strict=True)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m.to(DEV).set_precision("fp32")
    imgs = torch.randn(2, 3, 224, 224, generator=G(20))
    target = _dpt_inputs(21)[1]
    out = m(imgs.to(DEV))
    loss = ScaleAndShiftInvariantLoss(alpha=0.1)(out, target.to(DEV))
    loss.backward()
    # oracle: unmasked trunk with taps, then the DPT restatement
    cfg = mae_ref.MAEConfig(**{**mae_ref.VIT_B.__dict__})
    sdo = {k: v.clone().requires_grad_(v.is_floating_point() and "pos_embed" not in k) for k, v in sd.items()}
    taps = mae_ref.vit_trunk(sdo, cfg, imgs, dense=True)
    dsdo = {k[len("decoder."):]: v for k, v in sdo.items() if k.startswith("decoder.")}
    out_o = dpt_ref.dpt_forward(dsdo, taps)
    loss_o = dpt_ref.ssi_loss(out_o, target, alpha=0.1)
    loss_o.backward()
    assert rel_err(out, out_o) < 1e-3
    assert abs(float(loss.detach()) - float(loss_o)) < 1e-3 * abs(float(loss_o))
    for name in ("blocks.11.mlp.fc2.weight", "blocks.0.attn.qkv.weight", "patch_embed.proj.weight",
                 "decoder.layer1_rn.weight", "decoder.output_conv.0.weight"):
        p = dict(m.named_parameters())[name]
        assert rel_err(p.grad, sdo[name].grad) < 5e-3, name
    assert m.norm.weight.grad is None or float(m.norm.weight.grad.abs().max()) == 0.0  # final norm unused


def test_bilinear2x_two_by_two_kernels_equal_the_one_pixel_kernels_bit_for_bit(tmp_path):
    """the round-6 bilinear forward (2 x 2 outputs per thread, 9 taps instead of 16; the backward keeps one pixel per thread) against
    the one-pixel-per-thread kernels they replace (SSL4GIE_BILINEAR22=0, read once per process: a child process
    computes the reference) — identical bits, forward and backward, fp32 and bf16, odd sizes and the 1 x 1 map"""
    import subprocess
    import sys
    from ssl4gie_amd import ops
    shapes = [(2, 7, 5, 16), (3, 14, 14, 256), (2, 9, 33, 128), (1, 1, 1, 8), (2, 56, 56, 64), (1, 2, 3, 8)]
    code = (
        "import sys, torch\n"
        f"sys.path.insert(0, {repr(ROOT)})\n"
        "from ssl4gie_amd import ops\n"
        f"out = {{}}\n"
        f"for i, s in enumerate({shapes!r}):\n"
        "    for dt in (torch.float32, torch.bfloat16):\n"
        "        g = torch.Generator().manual_seed(100 + i)\n"
        "        x = torch.randn(*s, generator=g).to(dt).cuda()\n"
        "        y = ops.bilinear2x_fwd(x)\n"
        "        dy = torch.randn(y.shape, generator=g).to(dt).cuda()\n"
        "        out[(i, str(dt))] = (y.cpu(), ops.bilinear2x_bwd(dy).cpu())\n"
        f"torch.save(out, {repr(str(tmp_path / 'ref.pt'))})\n")
    env = dict(os.environ, SSL4GIE_BILINEAR22="0")
    subprocess.run([sys.executable, "-c", code], check=True, env=env, timeout=240)
    ref = torch.load(str(tmp_path / "ref.pt"))
    for i, s in enumerate(shapes):
        for dt in (torch.float32, torch.bfloat16):
            g = torch.Generator().manual_seed(100 + i)
            x = torch.randn(*s, generator=g).to(dt).cuda()
            y = ops.bilinear2x_fwd(x)
            dy = torch.randn(y.shape, generator=g).to(dt).cuda()
            dx = ops.bilinear2x_bwd(dy)
            ry, rdx = ref[(i, str(dt))]
            assert torch.equal(y.cpu(), ry), ("fwd", s, dt)
            assert torch.equal(dx.cpu(), rdx), ("bwd", s, dt)
