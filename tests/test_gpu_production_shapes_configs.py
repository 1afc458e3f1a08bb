"""GPU tests at the PRODUCTION shapes of BASELINE.json configs[2] (MoCo-v3 ResNet50, B = 256 per GPU), configs[3]
(ViT-B + DPT depth, B = 128) and configs[4] (Barlow Twins ViT-B, B = 512): the shape-selected code paths that the
small fixtures never reach — `nt256_pick_nj`, the K-split caps (256 splits over 802 816 pixels), the wide slab
reduction, the partial-tile TN kernel, the `TileG<...>` geometry choice of the direct convolutions, the gathered
convolutions with millions of output pixels — checked EXACTLY with small-integer operands (every product and every
partial sum is exact in bf16 / fp32, so any tile-assignment, split, tail or swizzle slip is a bit error) against an
fp64 reference built from rocBLAS products on the device (not from the engine's own kernels), and one full-size step
per config checked through size-independent properties (finite loss and gradients, BatchNorm running statistics
equal to the batch statistics of the activations, momentum encoder == EMA of the base encoder, deterministic loss).
Reference shapes: SURVEY Appendix B / C / D, i.e. torchvision Bottleneck via /root/reference Models/models.py:106-126,
Models/DPT_decoder.py:396-482, Models/moco_v3/moco/builder.py:75-96."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF, F32, F64 = torch.bfloat16, torch.float32, torch.float64


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ssl4gie_amd import _lib
    _lib.load()


def ints(shape, seed, lo=-1, hi=2):
    """small integers drawn on the device (hundreds of millions of them: a host generator would take minutes)"""
    g = torch.Generator(DEV).manual_seed(seed)
    return torch.randint(lo, hi, shape, generator=g, device=DEV, dtype=torch.int8).to(BF)


def mm64(a, b):
    """fp64 product of two integer-valued bf16 matrices in row chunks (an [M, K] fp64 copy of a 800k-row operand
    would be gigabytes)"""
    out = torch.empty(a.shape[0], b.shape[1], dtype=F64, device=a.device)
    step = 1 << 18
    b64 = b.double()
    for i in range(0, a.shape[0], step):
        out[i:i + step] = a[i:i + step].double() @ b64
    return out


def colstats_ref(y_bf, rows=128):
    """[ceil(M / rows), 2, N]: per-`rows`-row column sums and sums of squares of the STORED values"""
    M, N = y_bf.shape
    pad = (-M) % rows
    y = torch.cat([y_bf.double(), torch.zeros(pad, N, dtype=F64, device=y_bf.device)]) if pad else y_bf.double()
    y = y.view(-1, rows, N)
    return torch.stack([y.sum(1), (y * y).sum(1)], 1)


# ------------------------------------------------------------------------------------------------------------------
# configs[2]: the 1x1 convolutions of torchvision ResNet50 at B = 256: y[M, N] = x[M, K] W[N, K]^T
# with M = 256 * Hout * Wout pixels, bias-free, BatchNorm partial statistics in the epilogue; their data gradients
# (the transposed product) and weight gradients (contraction over the M pixels: one or two output tiles, up to 256
# K-splits, the wide slab reduction, the partial-tile TN kernel for N or M = 64 / 128)
R50_1X1 = [  # (M, N = Cout, K = Cin)
    (802816, 64, 64), (802816, 256, 64), (802816, 64, 256), (802816, 128, 256),        # layer1, layer2.0.conv1
    (200704, 512, 128), (200704, 128, 512), (200704, 256, 512), (200704, 512, 256),    # layer2 (+ downsample)
    (50176, 1024, 256), (50176, 256, 1024), (50176, 512, 1024), (50176, 1024, 512),    # layer3
    (12544, 2048, 512), (12544, 512, 2048), (12544, 2048, 1024),                       # layer4
]


@pytest.mark.parametrize("M,N,K", R50_1X1)
def test_r50_1x1_products_full_batch_exact(M, N, K):
    from ssl4gie_amd import ops
    x, w = ints((M, K), 301), ints((N, K), 302)
    ref = mm64(x, w.t())
    exact_bf = ref.to(BF)  # the fp32 accumulator holds the integer sum exactly: the stored value is its RNE rounding
    # forward with the BatchNorm partial statistics of the stored values (LinearFn want_stats)
    if ops.colstats_ok(M, N, K, BF):
        y, st = ops.linear_fwd(x, w, None, colstats=True)
        assert torch.equal(y, exact_bf), f"forward + colstats: max diff {(y.double() - ref).abs().max()}"
        assert torch.equal(st.double(), colstats_ref(exact_bf)), "BatchNorm partial statistics"
        st2 = ops.linear_colstats_only(x, w)  # the statistics-only product (C == NULL) of the no-grad path
        assert torch.equal(st2, st)
    y = ops.linear_fwd(x, w, None)
    assert torch.equal(y, exact_bf), "forward"
    del y
    # data gradient dX[M, K] = dY[M, N] W[N, K] on the transposed operand copy
    dy = ints((M, N), 303)
    dx = ops.linear_bwd_data(dy, w, w.t().contiguous())
    assert torch.equal(dx, mm64(dy, w).to(BF)), "data gradient"
    del dx
    # weight gradient dW[N, K] = dY^T X over M pixels (|sum| <= M < 2^24: exact whatever the split)
    refw = dy.double().t() @ x.double() if M <= 1 << 18 else sum(
        dy[i:i + (1 << 18)].double().t() @ x[i:i + (1 << 18)].double() for i in range(0, M, 1 << 18))
    dw = ops.linear_bwd_weight(dy, x)
    assert torch.equal(dw.double(), refw), f"weight gradient: max diff {(dw.double() - refw).abs().max()}"
    dw2 = ops.linear_bwd_weight(dy, x, out=dw.clone(), accumulate=True)
    assert torch.equal(dw2.double(), 2 * refw), "accumulating weight gradient"


# ------------------------------------------------------------------------------------------------------------------
# configs[3]: the 3x3 convolutions of the DPT depth decoder at B = 128 THROUGH the layer op
# (dpt_engine.Conv3x3Fn: direct / gathered / materialised path chosen per geometry, data gradient on the flipped
# weight, weight gradient + bias gradient), fp64 reference from nine shifted rocBLAS products
def conv_ref(x, w, stride):
    """x [B,H,W,Ci] (integer-valued bf16), w [Co,Ci,3,3] fp64 -> fp64 [B,Ho,Wo,Co], pad 1"""
    B, H, W, Ci = x.shape
    Co = w.shape[0]
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    xp = torch.zeros(B, H + 2, W + 2, Ci, dtype=BF, device=x.device)
    xp[:, 1:H + 1, 1:W + 1] = x
    out = torch.zeros(B * Ho * Wo, Co, dtype=F64, device=x.device)
    for dy in range(3):
        for dx in range(3):
            tap = xp[:, dy:dy + stride * (Ho - 1) + 1:stride, dx:dx + stride * (Wo - 1) + 1:stride].reshape(-1, Ci)
            out += mm64(tap, w[:, :, dy, dx].t().to(BF))
    return out.view(B, Ho, Wo, Co)


def conv_wgrad_ref(dy, x, stride):
    """dW[Co,Ci,3,3] fp64 = sum over output pixels of dy x patch"""
    B, H, W, Ci = x.shape
    _, Ho, Wo, Co = dy.shape
    xp = torch.zeros(B, H + 2, W + 2, Ci, dtype=BF, device=x.device)
    xp[:, 1:H + 1, 1:W + 1] = x
    dw = torch.zeros(Co, Ci, 3, 3, dtype=F64, device=x.device)
    d2 = dy.reshape(-1, Co)
    step = 1 << 18
    for ky in range(3):
        for kx in range(3):
            tap = xp[:, ky:ky + stride * (Ho - 1) + 1:stride, kx:kx + stride * (Wo - 1) + 1:stride].reshape(-1, Ci)
            acc = torch.zeros(Co, Ci, dtype=F64, device=x.device)
            for i in range(0, d2.shape[0], step):
                acc += d2[i:i + step].double().t() @ tap[i:i + step].double()
            dw[:, :, ky, kx] = acc
    return dw


DPT_CONVS = [  # (name, B, H, W, Cin, Cout, stride, relu_in, bias)
    ("refinenet1.resConfUnit.conv", 128, 56, 56, 256, 256, 1, True, True),
    ("refinenet2.resConfUnit.conv", 128, 28, 28, 256, 256, 1, True, True),
    ("refinenet3.resConfUnit.conv", 128, 14, 14, 256, 256, 1, True, True),
    ("layer1_rn", 128, 56, 56, 96, 256, 1, False, False),
    ("layer2_rn", 128, 28, 28, 192, 256, 1, False, False),
    ("layer4_rn", 128, 7, 7, 768, 256, 1, False, False),
    ("act_postprocess42.1", 128, 14, 14, 768, 768, 2, False, True),
    ("output_conv.0", 128, 112, 112, 256, 128, 1, False, True),
    ("output_conv.2", 128, 224, 224, 128, 32, 1, False, True),
]


@pytest.mark.parametrize("name,B,H,W,Ci,Co,stride,relu_in,bias", DPT_CONVS)
def test_dpt_conv3x3_layers_full_batch_exact(name, B, H, W, Ci, Co, stride, relu_in, bias):
    from ssl4gie_amd.dpt_engine import Conv3x3Fn
    from ssl4gie_amd.engine import GradSink, LPCache
    # operands in {0, 1} x {-1, 0, 1}: 9 Ci <= 6912 terms, |y| stays far below 2^24; bf16 outputs compared with the
    # RNE rounding of the exact value; the weight gradient sums B Ho Wo <= 6.4 M products of magnitude <= 1: exact
    x = ints((B, H, W, Ci), 311, 0, 2)
    w = ints((Co, Ci, 3, 3), 312).float()
    b = ints((Co,), 313, -2, 3).float() if bias else None
    xg = x.clone().requires_grad_(True)
    wg = w.clone().requires_grad_(True)
    bg = b.clone().requires_grad_(True) if bias else None
    y = Conv3x3Fn.apply(xg, wg, bg, stride, relu_in, GradSink(None), LPCache())
    ref = conv_ref(x, w.double(), stride)  # x >= 0: the ReLU in front is the identity on the values ...
    if bias:
        ref = ref + b.double()
    assert torch.equal(y.detach(), ref.to(BF)), f"{name} forward: max diff {(y.double() - ref).abs().max()}"
    del ref
    dy = ints(tuple(y.shape), 314)
    y.backward(dy)
    refw = conv_wgrad_ref(dy, x, stride)
    assert torch.equal(wg.grad.double(), refw), f"{name} weight gradient: max diff {(wg.grad.double() - refw).abs().max()}"
    del refw
    if bias:
        assert torch.equal(bg.grad.double(), dy.double().sum((0, 1, 2))), f"{name} bias gradient"
    # data gradient = full correlation of dy with the flipped kernel (stride 1) / its strided scatter (stride 2);
    # ... and its mask passes everything where x > 0, nothing where x == 0
    if stride == 1:
        refx = conv_ref(dy, w.double().flip(2, 3).transpose(0, 1).contiguous(), 1)
    else:
        up = torch.zeros(B, H, W, Co, dtype=BF, device=DEV)
        up[:, ::2, ::2] = dy
        refx = conv_ref(up, w.double().flip(2, 3).transpose(0, 1).contiguous(), 1)
    if relu_in:
        refx = refx * (x > 0)
    assert torch.equal(xg.grad, refx.to(BF)), f"{name} data gradient: max diff {(xg.grad.double() - refx).abs().max()}"


# ------------------------------------------------------------------------------------------------------------------
# configs[4]: the Barlow Twins projector (8192-8192-8192 on B = 512 rows per view) and the 8192 x 8192 x 512
# cross-correlation (a TN product with 1024 output tiles and 8 K-tiles: no split at all)
@pytest.mark.parametrize("M,N,K", [(512, 8192, 768), (512, 8192, 8192), (1024, 8192, 8192)])
def test_bt_projector_products_exact(M, N, K):
    from ssl4gie_amd import ops
    x, w = ints((M, K), 321), ints((N, K), 322)
    ref = mm64(x, w.t())
    assert torch.equal(ops.linear_fwd(x, w, None), ref.to(BF)), "forward"
    if ops.colstats_ok(M, N, K, BF):
        y, st = ops.linear_fwd(x, w, None, colstats=True)
        assert torch.equal(y, ref.to(BF)) and torch.equal(st.double(), colstats_ref(ref.to(BF))), "forward + statistics"
    dy = ints((M, N), 323)
    assert torch.equal(ops.linear_bwd_data(dy, w, w.t().contiguous()), mm64(dy, w).to(BF)), "data gradient"
    assert torch.equal(ops.linear_bwd_weight(dy, x).double(), dy.double().t() @ x.double()), "weight gradient"


def test_bt_cross_correlation_exact():
    from ssl4gie_amd import ops
    z1, z2 = ints((512, 8192), 331, -2, 3), ints((512, 8192), 332, -2, 3)
    c = ops.linear_bwd_weight(z1, z2)  # c[i, j] = sum_b z1[b, i] z2[b, j]
    assert torch.equal(c.double(), z1.double().t() @ z2.double())


# ------------------------------------------------------------------------------------------------------------------
# one full-size step per config, through size-independent properties
def _finite_grads(model):
    n = 0
    for name, p in model.named_parameters():
        if p.requires_grad and p.grad is not None:
            assert bool(torch.isfinite(p.grad).all()), name
            n += 1
    return n


def test_batchnorm_running_statistics_full_batch():
    """BatchNorm2d (+ ReLU) over the stem map of a 256-image batch (3.2 M rows x 64 channels): the running
    statistics after one training-mode call are momentum-blends of the batch statistics of the activations"""
    from ssl4gie_amd.engine import GradSink
    from ssl4gie_amd.resnet_engine import BatchNormFn
    rows, C = 256 * 112 * 112, 64
    g = torch.Generator(DEV).manual_seed(5)
    x = (torch.randn(rows, C, generator=g, device=DEV) * 1.5 + 0.3).to(BF)
    bn = torch.nn.BatchNorm2d(C).to(DEV)
    y = BatchNormFn.apply(x, bn.weight, bn.bias, None, bn, True, GradSink(None))
    xd = x.double()
    mean, var = xd.mean(0), xd.var(0, unbiased=True)
    assert torch.allclose(bn.running_mean.double(), 0.1 * mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(bn.running_var.double(), 0.9 + 0.1 * var, rtol=1e-5, atol=1e-6)
    yr = torch.relu((xd - mean) * torch.rsqrt(xd.var(0, unbiased=False) + bn.eps))
    assert (y.double() - yr).abs().max() <= 2 ** -7 * yr.abs().max()


def test_moco_r50_full_batch_step_properties():
    """BASELINE.json configs[2] at full size (B = 256 per GPU, bf16 engine, one GPU): one MoCo-v3 step"""
    from functools import partial
    from ssl4gie_amd.Models.moco_v3.moco import builder
    from ssl4gie_amd.Models.resnet import resnet50
    torch.manual_seed(0)
    model = builder.MoCo_ResNet(partial(resnet50, zero_init_residual=True), 256, 4096, 1.0).to(DEV).set_precision("bf16")
    g = torch.Generator("cpu").manual_seed(0)
    x1 = torch.randn(256, 3, 224, 224, generator=g).to(DEV)
    x2 = torch.randn(256, 3, 224, 224, generator=g).to(DEV)
    base0 = [p.detach().clone() for p in model.base_encoder.parameters()]
    mom0 = [p.detach().clone() for p in model.momentum_encoder.parameters()]
    rm0 = model.base_encoder.bn1.running_mean.clone()
    m = 0.99
    loss = model(x1, x2, m)
    assert torch.isfinite(loss) and 0.0 < float(loss) < 50.0
    loss.backward()
    assert _finite_grads(model) > 150
    # momentum encoder == EMA of the base encoder (builder.py:57-61), an exact fp32 axpby per element
    for pm, p0, pb in zip(model.momentum_encoder.parameters(), mom0, base0):
        assert torch.allclose(pm.detach(), p0 * m + pb * (1.0 - m), rtol=0, atol=1e-7 * float(pb.abs().max() + 1))
        assert pm.grad is None
    # the stem BatchNorm saw two views: its running mean moved twice towards the batch means of conv1's output
    bn1 = model.base_encoder.bn1
    assert bool(torch.isfinite(bn1.running_mean).all()) and bool(torch.isfinite(bn1.running_var).all())
    assert not torch.equal(bn1.running_mean, rm0)
    # deterministic: the same step on a fresh copy of the initial state gives the same loss
    torch.manual_seed(0)
    model2 = builder.MoCo_ResNet(partial(resnet50, zero_init_residual=True), 256, 4096, 1.0).to(DEV).set_precision("bf16")
    assert float(model2(x1, x2, m)) == float(loss)


def test_depth_vitb_dpt_full_batch_step_properties():
    """BASELINE.json configs[3] at full size (B = 128 per GPU, bf16 engine): ViT_from_MAE(dense='depth') + SSI loss"""
    from ssl4gie_amd.Models import models
    from ssl4gie_amd.losses import ScaleAndShiftInvariantLoss
    torch.manual_seed(0)
    model = models.ViT_from_MAE(None, False, 1, False, "depth", False, None, 768, 12, 12, "cls").to(DEV).set_precision("bf16")
    g = torch.Generator("cpu").manual_seed(0)
    imgs = torch.randn(128, 3, 224, 224, generator=g).to(DEV)
    tgt = torch.rand(128, 1, 224, 224, generator=g)
    tgt = torch.where(torch.rand(128, 1, 224, 224, generator=g) < 0.1, torch.zeros(()), tgt).to(DEV)
    loss_fn = ScaleAndShiftInvariantLoss(alpha=0.1)
    out = model(imgs)
    assert out.shape == (128, 1, 224, 224) and bool(torch.isfinite(out).all())
    assert float(out.min()) >= 0.0 and float(out.max()) <= 1.0  # Sigmoid head (DPT_decoder.py:481)
    loss = loss_fn(out, tgt)
    assert torch.isfinite(loss)
    loss.backward()
    n = _finite_grads(model)
    assert n > 200
    # the parameters the reference's graph never reaches stay without a gradient (norm.*,
    # decoder.refinenet4.resConfUnit1.*)
    for name, p in model.named_parameters():
        if name.startswith("norm.") or "refinenet4.resConfUnit1" in name:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
    with torch.no_grad():
        assert float(loss_fn(model(imgs), tgt)) == float(loss)


def test_barlow_twins_full_batch_step_properties():
    """BASELINE.json configs[4] at full size (two views of 512 images, bf16 engine)"""
    from ssl4gie_amd.Models.barlow_twins import BarlowTwins
    from ssl4gie_amd.Models.moco_v3 import vits
    torch.manual_seed(0)
    bb = vits.vit_base(num_classes=8)
    del bb.head
    model = BarlowTwins(bb, 768, "8192-8192-8192", lambd=0.0051).to(DEV).set_precision("bf16")
    g = torch.Generator("cpu").manual_seed(0)
    y1 = torch.randn(512, 3, 224, 224, generator=g).to(DEV)
    y2 = (y1.cpu() + 0.5 * torch.randn(512, 3, 224, 224, generator=g)).to(DEV)
    loss = model(y1, y2)
    assert torch.isfinite(loss) and float(loss) > 0
    loss.backward()
    assert _finite_grads(model) > 100
    with torch.no_grad():
        assert float(model(y1, y2)) == float(loss)
