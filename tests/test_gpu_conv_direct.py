"""GPU tests of the direct (halo-in-LDS) 3x3 convolution kernels behind the narrow end of the DPT
output head (reference: nn.Conv2d(128, 32, 3, 1, 1), Models/DPT_decoder.py:473-478, and its
gradients).  Operands are small integers, so every product and every partial sum is exact in
bf16 / fp32 and the comparison with torch's fp64 convolution on the CPU is BIT-EXACT: a wrong
halo offset, swizzle, tap shift, MFMA lane mapping or edge mask is an integer error.  Geometries
cover one to six 32-channel passes, one and two 32-cout blocks per workgroup, several cout groups,
maps that are not a multiple of the tile, all three tile geometries (8 x 32, 16 x 16, four images of 8 x 8),
single-row / single-column maps, and the production 224 x 224 map."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ssl4gie_amd import _lib
    _lib.load()


def ints(shape, seed, lo=-2, hi=3):
    g = torch.Generator("cpu").manual_seed(seed)
    return torch.randint(lo, hi, shape, generator=g).double()


def ref_conv(x, w, bias, relu_in):
    """x [B,H,W,Ci] fp64, w [Co,Ci,3,3] fp64 -> [B,H,W,Co]"""
    xi = x.permute(0, 3, 1, 2)
    if relu_in:
        xi = xi.clamp_min(0)
    return F.conv2d(xi, w, bias, padding=1).permute(0, 2, 3, 1).contiguous()


def w2_of(w):  # [Co,Ci,3,3] -> [Co, 9 Ci], taps row-major, channels innermost
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


GEOMS = [  # B, H, W, Cin, Cout
    (2, 16, 64, 128, 32),   # output_conv.2: four channel passes, whole tiles
    (1, 13, 37, 64, 32),    # ragged map
    (2, 9, 33, 32, 128),    # two 32-cout blocks per workgroup, two cout groups (data-gradient shape)
    (1, 1, 5, 32, 8),       # single row, fewer couts than one block
    (1, 40, 1, 64, 24),     # single column
    (3, 8, 32, 96, 40),     # three passes; Cout not a multiple of 32
    (1, 11, 40, 192, 128),  # output_conv.0-like: six passes, two cout groups of 64
    (3, 14, 14, 64, 64),    # 16 x 16 tiles (ResNet layer3 maps)
    (2, 16, 9, 32, 40),     # 16 x 16 tiles, ragged
    (9, 7, 7, 64, 128),     # 8 x 8 tiles of four images (layer4 maps); the last tile holds one image
    (5, 5, 8, 32, 32),      # 8 x 8 tiles, ragged rows
]


@pytest.mark.parametrize("B,H,W,Ci,Co", GEOMS)
@pytest.mark.parametrize("relu_in", [False, True])
def test_direct_conv_forward_exact(B, H, W, Ci, Co, relu_in):
    from ssl4gie_amd import _lib, ops
    assert _lib.load().ssl4gie_conv3x3_direct_ok(B, H, W, Ci, Co)
    x = ints((B, H, W, Ci), 11)
    w = ints((Co, Ci, 3, 3), 12, -1, 2)
    bias = ints((Co,), 13)
    ref = ref_conv(x, w, bias, relu_in)
    y = ops.conv3x3_direct_fwd(x.to(DEV, BF), w2_of(w).to(DEV, BF), bias.float().to(DEV), relu=relu_in)
    assert ref.abs().max() < 256  # exactly representable in bf16
    assert torch.equal(y.double().cpu(), ref)


@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 16, 64, 64, 64), (1, 13, 37, 64, 32), (2, 9, 33, 128, 128),
                                         (1, 5, 40, 32, 40), (3, 14, 14, 64, 64), (9, 7, 7, 64, 64)])
def test_direct_conv_batchnorm_partials_exact(B, H, W, Ci, Co):
    """colstats: per-tile column sums / sums of squares of the stored outputs (what BatchNormFn takes
    from the producing convolution); ragged tiles must not count their overhang"""
    from ssl4gie_amd import _lib, ops
    x = ints((B, H, W, Ci), 41)
    w = ints((Co, Ci, 3, 3), 42, -1, 2)
    ref = ref_conv(x, w, None, False)
    y, stats = ops.conv3x3_direct_fwd(x.to(DEV, BF), w2_of(w).to(DEV, BF), None, colstats=True)
    assert torch.equal(y.double().cpu(), ref)
    assert stats.shape == (_lib.load().ssl4gie_conv3x3_direct_tiles(B, H, W), 2, Co)
    tot = stats.double().sum(0).cpu()
    assert torch.equal(tot[0], ref.sum((0, 1, 2))) and torch.equal(tot[1], (ref * ref).sum((0, 1, 2)))


@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 9, 33, 32, 128), (1, 16, 64, 128, 32)])
def test_direct_conv_relu_mask_epilogue_exact(B, H, W, Ci, Co):
    """the data gradient of a convolution whose input went through a ReLU: y = mask > 0 ? y : 0"""
    from ssl4gie_amd import ops
    x = ints((B, H, W, Ci), 21)
    w = ints((Co, Ci, 3, 3), 22, -1, 2)
    mask = ints((B, H, W, Co), 23)
    ref = ref_conv(x, w, None, False) * (mask > 0)
    y = ops.conv3x3_direct_fwd(x.to(DEV, BF), w2_of(w).to(DEV, BF), None, relu_mask=mask.to(DEV, BF))
    assert torch.equal(y.double().cpu(), ref)


@pytest.mark.parametrize("B,H,W,Ci,Co", [(2, 16, 64, 128, 32), (1, 13, 37, 64, 32), (3, 5, 100, 192, 32),
                                         (1, 1, 3, 64, 32), (2, 9, 40, 128, 128), (1, 17, 33, 64, 96),
                                         (3, 14, 14, 64, 64), (9, 7, 7, 128, 32), (5, 5, 8, 64, 64),
                                         (2, 16, 9, 64, 32)])
@pytest.mark.parametrize("relu_in", [False, True])
def test_direct_conv_weight_gradient_exact(B, H, W, Ci, Co, relu_in):
    from ssl4gie_amd import ops
    x = ints((B, H, W, Ci), 31)
    dy = ints((B, H, W, Co), 32, -1, 2)
    xr = x.clamp_min(0) if relu_in else x
    # dW[co, ci, ky, kx] = sum_p dy[p, co] x[p + (ky - 1, kx - 1), ci]
    ref = torch.nn.grad.conv2d_weight(xr.permute(0, 3, 1, 2), (Co, Ci, 3, 3), dy.permute(0, 3, 1, 2),
                                      padding=1)
    db = torch.full((Co,), 7.0, device=DEV)
    dw2 = ops.conv3x3_direct_wgrad(dy.to(DEV, BF), x.to(DEV, BF), relu=relu_in, bias_out=db)
    assert ref.abs().max() < 2 ** 24
    assert torch.equal(dw2.double().cpu(), w2_of(ref))
    assert torch.equal(db.double().cpu(), dy.sum((0, 1, 2)))


def test_direct_conv_wide_channels_small_maps_exact():
    """ResNet layer4 geometry (512 -> 512 on 7 x 7 maps): sixteen channel passes, eight cout groups,
    8 x 8 tiles of four images; weight gradient with 8 x 16 (slice, group) combinations"""
    from ssl4gie_amd import ops
    B, H, W, Ci, Co = 6, 7, 7, 512, 512
    x = ints((B, H, W, Ci), 51, -1, 2)
    w = ints((Co, Ci, 3, 3), 52, -1, 2)
    ref = ref_conv(x, w, None, False)
    assert ref.abs().max() < 256
    y = ops.conv3x3_direct_fwd(x.to(DEV, BF), w2_of(w).to(DEV, BF), None)
    assert torch.equal(y.double().cpu(), ref)
    dy = ints((B, H, W, Co), 53, -1, 2)
    refw = torch.nn.grad.conv2d_weight(x.permute(0, 3, 1, 2), (Co, Ci, 3, 3), dy.permute(0, 3, 1, 2), padding=1)
    db = torch.zeros(Co, device=DEV)
    dw2 = ops.conv3x3_direct_wgrad(dy.to(DEV, BF), x.to(DEV, BF), bias_out=db)
    assert torch.equal(dw2.double().cpu(), w2_of(refw))
    assert torch.equal(db.double().cpu(), dy.sum((0, 1, 2)))


def test_direct_conv_matches_the_gathered_gemm_at_the_production_head_shape():
    """output_conv.2 at the depth-finetune geometry (224 x 224, 128 -> 32), random bf16 operands:
    forward against the gathered 256x256 GEMM path (same bf16 inputs, fp32 accumulation: only the
    summation order differs), weight gradient against its TN kernel, data gradient against the
    materialised-patch path."""
    from ssl4gie_amd import ops
    B, H, W, Ci, Co = 4, 224, 224, 128, 32
    g = torch.Generator("cpu").manual_seed(5)
    x = torch.randn(B, H, W, Ci, generator=g).to(DEV, BF)
    w2 = (torch.randn(Co, 9 * Ci, generator=g) * 0.03).to(DEV, BF)
    bias = torch.randn(Co, generator=g).to(DEV)
    dy = torch.randn(B, H, W, Co, generator=g).to(DEV, BF)
    y_ref = ops.conv3x3_fwd(x, w2, bias, 1, True)
    y = ops.conv3x3_direct_fwd(x, w2, bias, relu=True)
    assert (y.float() - y_ref.float()).abs().max() <= 2e-2 * y_ref.float().abs().max()
    assert ((y.float() - y_ref.float()).norm() / y_ref.float().norm()) < 3e-3
    dw_ref = ops.conv3x3_bwd_weight(dy.view(-1, Co), x, 1, True)
    dw = ops.conv3x3_direct_wgrad(dy, x, relu=True)
    assert ((dw - dw_ref).norm() / dw_ref.norm()) < 1e-4
    wd = (torch.randn(Ci, 9 * Co, generator=g) * 0.05).to(DEV, BF)
    dx = ops.conv3x3_direct_fwd(dy, wd, None, relu_mask=x)
    ld2 = ops.k_pad(9 * Co, BF)
    wdp = torch.zeros(Ci, ld2, device=DEV, dtype=BF)
    wdp[:, :9 * Co] = wd
    dx_ref = ops.relu_bwd(x, ops.linear_fwd(ops.im2col3x3(dy, 1, False, ld2), wdp, None,
                                            out_dtype=BF).view(B, H, W, Ci))
    assert ((dx.float() - dx_ref.float()).norm() / dx_ref.float().norm()) < 3e-3
    assert torch.equal(dx == 0, dx_ref == 0) or ((dx == 0) != (dx_ref == 0)).float().mean() < 1e-4


@pytest.mark.parametrize("B,H,W", [(2, 32, 64), (3, 37, 45), (1, 7, 9), (2, 224, 224), (5, 33, 32)])
def test_direct_stem_forward_and_weight_gradient_exact(B, H, W):
    """7x7 stride-2 pad-3 stem (torchvision conv1) on the direct kernels: packed image, forward with
    BatchNorm partial statistics, weight gradient — integers, exact against torch fp64"""
    from ssl4gie_amd import _lib, ops
    x = ints((B, 3, H, W), 61)
    w = ints((64, 3, 7, 7), 62, -1, 2)
    ref = F.conv2d(x, w, None, stride=2, padding=3).permute(0, 2, 3, 1).contiguous()
    Ho, Wo = ref.shape[1:3]
    assert ref.abs().max() < 256
    packed = ops.stem7x7_pack(x.float().to(DEV))
    w2s = ops.stem7x7_weight(w.float().to(DEV)).to(BF)
    y, stats = ops.stem7x7_fwd(packed, w2s, B, H, W, colstats=True)
    assert y.shape == (B, Ho, Wo, 64)
    assert torch.equal(y.double().cpu(), ref)
    tot = stats.double().sum(0).cpu()
    assert torch.equal(tot[0], ref.sum((0, 1, 2))) and torch.equal(tot[1], (ref * ref).sum((0, 1, 2)))
    dy = ints((B, Ho, Wo, 64), 63, -1, 2)
    refw = torch.nn.grad.conv2d_weight(x, (64, 3, 7, 7), dy.permute(0, 3, 1, 2), stride=2, padding=3)
    dw = ops.stem7x7_wgrad(dy.to(DEV, BF), packed, B, H, W)
    assert refw.abs().max() < 2 ** 24
    assert torch.equal(dw.double().cpu(), refw)
