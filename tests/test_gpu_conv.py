"""GPU parity of the implicit 3x3 convolution (ssl4gie_gemm_desc.conv: the patch matrix gathered
inside the 256x256 GEMM kernels) against torch fp32 conv2d on the same bf16-rounded inputs and
against the materialised-patch-matrix path of the same library."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ssl4gie_amd import _lib
    _lib.load()


def G(seed):
    return torch.Generator("cpu").manual_seed(seed)


def _case(B, H, W, Cin, Cout, seed):
    x = torch.randn(B, H, W, Cin, generator=G(seed)).to(BF)
    w = (torch.randn(Cout, Cin, 3, 3, generator=G(seed + 1)) * (9 * Cin) ** -0.5).to(BF)
    b = torch.randn(Cout, generator=G(seed + 2))
    return x, w, b


def _w2(w):
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


FWD_CASES = [
    # B, H, W, Cin, Cout, stride, relu
    (2, 14, 14, 64, 64, 1, False),
    (2, 14, 14, 128, 96, 1, True),    # ragged N tile, two K-tiles per tap
    (3, 9, 7, 64, 40, 2, True),       # odd map, stride 2, ragged everything
    (1, 28, 28, 256, 256, 1, False),  # 4 row tiles, the last one ragged
    (5, 20, 24, 64, 264, 1, True),    # two column tiles, 10 row tiles (persistent workgroups)
    (2, 2, 3, 64, 8, 1, False),       # smallest map: every pixel touches the border
]


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,relu", FWD_CASES)
def test_conv3x3_implicit_forward(B, H, W, Cin, Cout, stride, relu):
    from ssl4gie_amd import ops
    x, w, b = _case(B, H, W, Cin, Cout, 10)
    xd, wd, bd = x.to(DEV), _w2(w).to(DEV), b.to(DEV)
    assert ops.conv3x3_implicit_ok(xd, stride, Cout)
    y = ops.conv3x3_fwd(xd, wd, bd, stride, relu).float().cpu()
    xin = F.relu(x.float()) if relu else x.float()
    ref = F.conv2d(xin.permute(0, 3, 1, 2), w.float(), b, stride=stride, padding=1).permute(0, 2, 3, 1)
    assert y.shape == ref.shape
    assert rel_err(y, ref) < 4e-3  # bf16 output rounding
    # same product through the materialised patch matrix: identical operands, fp32 accumulation
    cols = ops.im2col3x3(xd, stride, relu)
    y2 = ops.linear_fwd(cols, wd, bd).float().cpu().view_as(y)
    assert rel_err(y, y2) < 2e-3
    # no bias: the plain epilogue
    y3 = ops.conv3x3_fwd(xd, wd, None, stride, relu).float().cpu()
    assert rel_err(y3, ref - b) < 4e-3


WGRAD_CASES = [
    # B, H, W, Cin, Cout, stride, relu
    (4, 8, 8, 64, 64, 1, False),
    (2, 16, 16, 128, 96, 1, True),
    (4, 16, 16, 64, 40, 2, True),
    (64, 7, 7, 64, 64, 1, False),    # 49-pixel maps: K-tiles straddle images
    (8, 16, 8, 32, 264, 1, True),    # taps narrower than a K-chunk group, two row tiles of dW
    (2, 32, 32, 24, 16, 1, False),   # C = 24: 256-column tiles cut through taps
]


@pytest.mark.parametrize("B,H,W,Cin,Cout,stride,relu", WGRAD_CASES)
def test_conv3x3_implicit_weight_gradient(B, H, W, Cin, Cout, stride, relu):
    from ssl4gie_amd import ops
    x, w, _ = _case(B, H, W, Cin, Cout, 20)
    Ho, Wo = ops.conv_out_hw(H, W, stride)
    dy = torch.randn(B, Ho, Wo, Cout, generator=G(23)).to(BF)
    xd, dyd = x.to(DEV), dy.to(DEV).view(-1, Cout)
    assert ops.conv3x3_implicit_ok(xd, stride, Cout, wgrad=True)
    db = torch.empty(Cout, device=DEV)
    dw2 = ops.conv3x3_bwd_weight(dyd, xd, stride, relu, bias_out=db).cpu()
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(False)
    xin = F.relu(xr) if relu else xr
    wr = w.float().requires_grad_(True)
    yr = F.conv2d(xin, wr, None, stride=stride, padding=1)
    yr.backward(dy.float().permute(0, 3, 1, 2))
    ref = wr.grad.permute(0, 2, 3, 1).reshape(Cout, -1)
    assert rel_err(dw2, ref) < 1e-4  # fp32 accumulation of exact bf16 products
    assert rel_err(db.cpu(), dy.float().sum((0, 1, 2))) < 1e-4
    dw3 = ops.conv3x3_bwd_weight(dyd, xd, stride, relu).cpu()  # without the fused bias gradient
    assert torch.equal(dw3, dw2)


def test_conv3x3_relu_mask_epilogue():
    """data gradient of conv(relu(x)): the mask x > 0 applied in the GEMM epilogue"""
    from ssl4gie_amd import ops
    x, w, _ = _case(3, 20, 24, 64, 128, 40)           # x: the convolution's input (mask source)
    dy = torch.randn(3, 20, 24, 128, generator=G(43)).to(BF)
    wflip = w.flip(2, 3).permute(1, 2, 3, 0).reshape(64, 9 * 128).contiguous()  # [Cin, 9 Cout]
    xd, dyd, wd = x.to(DEV), dy.to(DEV), wflip.to(DEV)
    dx = ops.conv3x3_fwd(dyd, wd, None, 1, False, relu_mask=xd).float().cpu()
    plain = ops.conv3x3_fwd(dyd, wd, None, 1, False).float().cpu()
    assert torch.equal(dx, torch.where(x.float() > 0, plain, torch.zeros(())))
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    F.conv2d(F.relu(xr), w.float(), None, padding=1).backward(dy.float().permute(0, 3, 1, 2))
    assert rel_err(dx, xr.grad.permute(0, 2, 3, 1)) < 4e-3


def test_conv3x3_fn_matches_materialised_path(monkeypatch):
    """Conv3x3Fn forward + backward: implicit path vs SSL4GIE_IMPLICIT_CONV=0 on one layer"""
    from ssl4gie_amd import dpt_engine
    from ssl4gie_amd.engine import GradSink, LPCache
    x, w, b = _case(4, 16, 16, 64, 128, 30)
    dy = torch.randn(4, 16, 16, 128, generator=G(33)).to(BF).to(DEV)
    out = {}
    for flag in (True, False):
        monkeypatch.setattr(dpt_engine, "_IMPLICIT", flag)
        wp = torch.nn.Parameter(w.float().to(DEV))
        bp = torch.nn.Parameter(b.to(DEV))
        xp = x.to(DEV).requires_grad_(True)
        y = dpt_engine.Conv3x3Fn.apply(xp, wp, bp, 1, True, GradSink(None), LPCache())
        y.backward(dy)
        out[flag] = (y.detach().float(), xp.grad.float(), wp.grad, bp.grad)
    for a, c in zip(out[True], out[False]):
        assert rel_err(a, c) < 3e-3


def test_conv_descriptor_rejected_when_geometry_does_not_fit():
    """a descriptor carrying `conv` never falls back silently: EARG for C % 64 != 0 on the NT side"""
    from ssl4gie_amd import _lib, ops
    x = torch.zeros(2, 8, 8, 32, dtype=BF, device=DEV)
    w2 = torch.zeros(16, 9 * 32, dtype=BF, device=DEV)
    assert not ops.conv3x3_implicit_ok(x, 1, 16)
    with pytest.raises(AssertionError):
        ops.conv3x3_fwd(x, w2)
    import ctypes as C
    d = ops._desc(2 * 8 * 8, 16, 9 * 32, _lib.BF16, _lib.BF16)
    g = ops._geom(x, 1, False)
    d.conv = C.pointer(g)
    y = torch.empty(128, 16, dtype=BF, device=DEV)
    d.A, d.sAm, d.sAk = ops.ptr(x), 288, 1
    d.B, d.sBk, d.sBn = ops.ptr(w2), 1, 288
    d.C, d.ldc = ops.ptr(y), 16
    rc = _lib.load().ssl4gie_gemm(C.byref(d), None, 0, ops.stream())
    assert rc == 1000


@pytest.mark.parametrize("cout,cin", [(32, 128), (96, 96), (64, 40)])
def test_conv3x3_weight_pack_equals_the_torch_relayouts(cout, cin):
    """ssl4gie_conv3x3_weight_pack (one launch, cast included) == the permute / flip / pad / cast chains it
    replaced, bit for bit, in the three layouts and both operand types"""
    from ssl4gie_amd import ops
    w = torch.randn(cout, cin, 3, 3, generator=torch.Generator().manual_seed(cout + cin)).to(DEV)
    for dt in (torch.bfloat16, torch.float32):
        for ld_extra in (0, 24):
            ld0, ld1 = 9 * cin + ld_extra, 9 * cout + ld_extra
            ref0 = torch.zeros(cout, ld0, device=DEV)
            ref0[:, :9 * cin] = w.permute(0, 2, 3, 1).reshape(cout, 9 * cin)
            ref1 = torch.zeros(cin, ld1, device=DEV)
            ref1[:, :9 * cout] = w.flip(2, 3).permute(1, 2, 3, 0).reshape(cin, 9 * cout)
            assert torch.equal(ops.conv3x3_weight_pack(w, dt, 0, ld0), ref0.to(dt))
            assert torch.equal(ops.conv3x3_weight_pack(w, dt, 1, ld1), ref1.to(dt))
            assert torch.equal(ops.conv3x3_weight_pack(w, dt, 2, ld0), ref0.to(dt).t().contiguous())


def test_conv3x3_wgrad_unpack_equals_the_torch_scatter():
    from ssl4gie_amd import ops
    cout, cin, ld = 48, 40, 9 * 40 + 24
    g = torch.Generator().manual_seed(2)
    dw2 = torch.randn(cout, ld, generator=g).to(DEV)
    base = torch.randn(cout, cin, 3, 3, generator=g).to(DEV)
    ref = dw2[:, :9 * cin].view(cout, 3, 3, cin).permute(0, 3, 1, 2)
    t = torch.empty_like(base)
    assert torch.equal(ops.conv3x3_wgrad_unpack(dw2, t, False), ref.contiguous())
    t = base.clone()
    assert torch.equal(ops.conv3x3_wgrad_unpack(dw2, t, True), base + ref)
