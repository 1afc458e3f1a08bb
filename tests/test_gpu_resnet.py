"""GPU parity of the ResNet50 trunk: glue kernels against torch fp32 of the same
op and the whole trunk (training-mode BatchNorm) against the CPU restatement.  torchvision is absent
from the reference checkout and this image, so parity at that boundary is unpinned (oracle header)."""
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


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("relu,with_res", [(False, False), (True, False), (True, True)])
def test_batchnorm_fwd_bwd_fp32(relu, with_res):
    from ssl4gie_amd.engine import GradSink
    from ssl4gie_amd.resnet_engine import BatchNormFn
    C = 64
    x = torch.randn(3, C, 9, 7, generator=G(1)) * 2 + 5  # |mean| >> std exercises the pivoted sums
    res = torch.randn(3, C, 9, 7, generator=G(2)) if with_res else None
    bn_ref = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn_ref.weight.copy_(torch.randn(C, generator=G(3)))
        bn_ref.bias.copy_(torch.randn(C, generator=G(4)))
    bn = torch.nn.BatchNorm2d(C)
    bn.load_state_dict(bn_ref.state_dict())
    bn.to(DEV)
    xr = x.clone().requires_grad_(True)
    rr = res.clone().requires_grad_(True) if with_res else None
    yr = bn_ref(xr)
    if with_res:
        yr = yr + rr
    if relu:
        yr = F.relu(yr)
    dy = torch.randn(yr.shape, generator=G(5))
    yr.backward(dy)
    xd = nhwc(x).to(DEV).requires_grad_(True)
    rd = nhwc(res).to(DEV).requires_grad_(True) if with_res else None
    y = BatchNormFn.apply(xd, bn.weight, bn.bias, rd, bn, relu, GradSink(None))
    y.backward(nhwc(dy).to(DEV))
    assert rel_err(nchw(y.detach().cpu()), yr) < 1e-5
    assert rel_err(nchw(xd.grad.cpu()), xr.grad) < 1e-4
    assert rel_err(bn.weight.grad.cpu(), bn_ref.weight.grad) < 1e-4
    assert rel_err(bn.bias.grad.cpu(), bn_ref.bias.grad) < 1e-4
    if with_res:
        assert rel_err(nchw(rd.grad.cpu()), rr.grad) < 1e-5
    assert rel_err(bn.running_mean.cpu(), bn_ref.running_mean) < 1e-5
    assert rel_err(bn.running_var.cpu(), bn_ref.running_var) < 1e-5
    assert int(bn.state_dict()["num_batches_tracked"]) == 1  # host-side count, written at state_dict()


@pytest.mark.parametrize("dtype", [F32, BF])
@pytest.mark.parametrize("rows,C", [(1531, 64), (4099, 24), (777, 1024)])
def test_batchnorm_relu_backward_mask_from_x_equals_mask_from_output(rows, C, dtype):
    """ssl4gie_bn_bwd_xmask (mask rebuilt from x and the forward's coefficients) against ssl4gie_bn_bwd reading
    the ReLU output: dx, dgamma and dbeta bit-identical, negative gammas and exact-zero pre-activations included"""
    from ssl4gie_amd import ops
    g = G(rows + C)
    x = (torch.randn(rows, C, generator=g) * 2 + 1).to(dtype).to(DEV)
    gamma = torch.randn(C, generator=g).to(DEV)
    beta = (torch.randn(C, generator=g) * 0.3).to(DEV)
    gamma[1] = 0.0                       # y = beta exactly: the whole channel on one side of the mask
    beta[1] = 0.0                        # ... at exactly zero: masked off on both paths
    gamma[2], beta[2] = 0.0, 0.25
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    y, mean, rstd = ops.bn_fwd(x, gamma, beta, None, rm, rv, 0.1, 1e-5, True, True)
    assert float((y[:, 1].float() != 0).sum()) == 0 and float((y[:, 2].float() <= 0).sum()) == 0
    dy = torch.randn(rows, C, generator=g).to(dtype).to(DEV)
    dg0, db0 = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    dg1, db1 = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    dx0, _ = ops.bn_bwd(dy, y, x, gamma, mean, rstd, True, False, dg0, db0, False)
    dx1 = ops.bn_bwd_xmask(dy, x, gamma, beta, mean, rstd, dg1, db1, False)
    assert torch.equal(dx0, dx1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)
    # accumulate
    dx2 = ops.bn_bwd_xmask(dy, x, gamma, beta, mean, rstd, dg1, db1, True)
    assert torch.equal(dx2, dx1) and torch.equal(dg1, 2 * dg0) and torch.equal(db1, 2 * db0)
    # the SyncBatchNorm halves (local sums -> [exchange] -> apply): mask from x == mask from the output
    s0, _ = ops.bn_bwd_reduce(dy, y, x, mean, rstd, True, False)
    s1 = ops.bn_bwd_reduce_xmask(dy, x, gamma, beta, mean, rstd)
    assert torch.equal(s0, s1)
    a0 = ops.bn_bwd_apply(dy, y, x, gamma, mean, rstd, s0, 1.0 / rows, True)
    a1 = ops.bn_bwd_apply_xmask(dy, x, gamma, beta, mean, rstd, s1, 1.0 / rows)
    assert torch.equal(a0, a1)


@pytest.mark.parametrize("dtype", [F32, BF])
def test_stem_bn_relu_maxpool_in_one_pass_equals_the_three_kernels(dtype):
    """BnReluMaxPoolFn (statistics from the convolution's partials, normalisation + ReLU inside the pool's window
    reads) against BatchNormFn + MaxPoolFn: pooled map, input gradient, dgamma / dbeta and running statistics
    bit-identical (odd map sizes: clipped windows; ReLU ties at 0)"""
    from ssl4gie_amd.engine import GradSink
    from ssl4gie_amd.resnet_engine import BatchNormFn, BnReluMaxPoolFn, MaxPoolFn
    B, H, W, C = 3, 17, 19, 64
    g = G(77)
    x = (torch.randn(B, H, W, C, generator=g) * 1.5 + 0.3).to(dtype).to(DEV)
    x2 = x.view(-1, C).float()
    pad = (-x2.shape[0]) % 128
    xp = torch.cat([x2, x2.new_zeros(pad, C)]).view(-1, 128, C)
    st = torch.stack([xp.sum(1), (xp * xp).sum(1)], 1).contiguous()     # what the stem convolution's epilogue emits
    outs = []
    for fused in (False, True):
        bn = torch.nn.BatchNorm2d(C).to(DEV)
        with torch.no_grad():
            bn.weight.copy_(torch.randn(C, generator=G(78)).to(DEV))
            bn.bias.copy_((0.3 * torch.randn(C, generator=G(79))).to(DEV))
        xi = x.clone().requires_grad_(True)
        if fused:
            y = BnReluMaxPoolFn.apply(xi, bn.weight, bn.bias, bn, GradSink(None), st)
        else:
            y = MaxPoolFn.apply(BatchNormFn.apply(xi, bn.weight, bn.bias, None, bn, True, GradSink(None), st))
        dy = torch.randn(y.shape, generator=G(80)).to(dtype).to(DEV)
        y.backward(dy)
        outs.append((y.detach(), xi.grad, bn.weight.grad, bn.bias.grad, bn.running_mean.clone(), bn.running_var.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert outs[0][0].shape == (B, 9, 10, C)


@pytest.mark.parametrize("B,H,W,C,grad", [(2, 24, 40, 64, True), (3, 28, 28, 128, True), (5, 14, 14, 256, False),
                                           (6, 7, 7, 512, False), (1, 9, 33, 64, True)])
def test_bn_relu_inside_the_direct_convolution_equals_the_separate_pass(B, H, W, C, grad):
    """BnReluConv3x3Fn (bn1 -> relu applied in the 3x3 direct kernels' halo staging, forward and weight gradient;
    mask-from-x BatchNorm backward) against BatchNormFn + Conv3x3Fn: output, its BatchNorm statistics, input
    gradient, dW, dgamma, dbeta, running statistics — bit-identical; padding stays zero AFTER the normalisation
    (maps with clipped tiles, all three tile geometries)"""
    from ssl4gie_amd.dpt_engine import Conv3x3Fn
    from ssl4gie_amd.engine import GradSink, LPCache
    from ssl4gie_amd import resnet_engine
    from ssl4gie_amd.resnet_engine import BatchNormFn, BnReluConv3x3Fn, bn_relu_conv3x3_ok
    resnet_engine._BN_CONV_FUSED = True   # opt-in path (SSL4GIE_BN_CONV_FUSED=1): measured null, kept correct
    g = G(90 + C)
    x = (torch.randn(B, H, W, C, generator=g) * 1.3 + 0.4).to(BF).to(DEV)
    x2 = x.view(-1, C).float()
    xp = torch.cat([x2, x2.new_zeros((-x2.shape[0]) % 128, C)]).view(-1, 128, C)
    st = torch.stack([xp.sum(1), (xp * xp).sum(1)], 1).contiguous()
    conv = torch.nn.Conv2d(C, C, 3, 1, 1, bias=False).to(DEV)
    outs = []
    for fused in (False, True):
        bn = torch.nn.BatchNorm2d(C).to(DEV)
        with torch.no_grad():
            bn.weight.copy_((1 + 0.3 * torch.randn(C, generator=G(91))).to(DEV))
            bn.bias.copy_((0.3 * torch.randn(C, generator=G(92))).to(DEV))
        conv.weight.grad = None
        lp = LPCache()
        xi = x.clone().requires_grad_(grad)
        with torch.set_grad_enabled(grad):
            if fused:
                assert bn_relu_conv3x3_ok(xi, bn, st, conv, grad) == (C <= 128)   # wider layers keep the separate pass
                y, ys = BnReluConv3x3Fn.apply(xi, bn.weight, bn.bias, bn, st, conv.weight, GradSink(None), lp, True)
            else:
                z = BatchNormFn.apply(xi, bn.weight, bn.bias, None, bn, True, GradSink(None), st)
                y, ys = Conv3x3Fn.apply(z, conv.weight, None, 1, False, GradSink(None), lp, True)
        res = [y.detach(), ys, bn.running_mean.clone(), bn.running_var.clone()]
        if grad:
            y.backward(torch.randn(y.shape, generator=G(93)).to(BF).to(DEV))
            res += [xi.grad, conv.weight.grad.clone(), bn.weight.grad, bn.bias.grad]
        outs.append(res)
    for i, (a, b) in enumerate(zip(*outs)):
        assert torch.equal(a, b), i
    # the normalised map's zero padding: a border output differs from what padding the RAW map would give
    assert float(outs[0][0].float().abs().sum()) > 0
    resnet_engine._BN_CONV_FUSED = False


@pytest.mark.parametrize("which", ["bn", "stem_pool", "bn_conv"])
def test_optimizer_step_between_forward_and_backward_is_refused_only_for_its_own_parameters(which):
    """The mask-from-x backwards rebuild the ReLU mask from gamma / beta as they are at backward time:
    a step of the optimizer that OWNS them between forward and backward must raise in all three autograd nodes
    (the fused optimizers do not bump Tensor._version), a step of an unrelated optimizer must not."""
    from ssl4gie_amd.dpt_engine import Conv3x3Fn  # noqa: F401
    from ssl4gie_amd.engine import GradSink, LPCache, bump_weights_epoch
    from ssl4gie_amd.resnet_engine import BatchNormFn, BnReluConv3x3Fn, BnReluMaxPoolFn
    B, H, W, C = 2, 16, 16, 64
    x = (torch.randn(B, H, W, C, generator=G(5)) * 1.3 + 0.2).to(BF).to(DEV)
    x2 = x.view(-1, C).float()
    xp = torch.cat([x2, x2.new_zeros((-x2.shape[0]) % 128, C)]).view(-1, 128, C)
    st = torch.stack([xp.sum(1), (xp * xp).sum(1)], 1).contiguous()
    conv = torch.nn.Conv2d(C, C, 3, 1, 1, bias=False).to(DEV)
    other = torch.nn.Parameter(torch.zeros(4, device=DEV))
    other.grad = torch.ones_like(other)

    def run(opt_of):
        bn = torch.nn.BatchNorm2d(C).to(DEV)
        xi = x.clone().requires_grad_(True)
        if which == "bn":
            y = BatchNormFn.apply(xi, bn.weight, bn.bias, None, bn, True, GradSink(None), st)
        elif which == "stem_pool":
            y = BnReluMaxPoolFn.apply(xi, bn.weight, bn.bias, bn, GradSink(None), st)
        else:
            y, _ = BnReluConv3x3Fn.apply(xi, bn.weight, bn.bias, bn, st, conv.weight, GradSink(None), LPCache(), False)
        # what ssl4gie_amd.optim's arena optimizers do: rewrite the storage through a raw pointer (no
        # Tensor._version bump, so autograd's own saved-tensor check cannot see it) and advance the weights epoch
        owned = opt_of(bn)
        with torch.no_grad():
            for q in owned:
                q.data.add_(0.25)
        bump_weights_epoch(touched=owned)
        y.backward(torch.ones_like(y))
        return xi.grad

    assert run(lambda bn: [other]) is not None            # somebody else's step: the backward runs
    with pytest.raises(RuntimeError, match="updated between"):
        run(lambda bn: [bn.weight, bn.bias])              # gamma / beta rewritten: refused


@pytest.mark.parametrize("rows,C", [(1531, 64), (4099, 256), (777, 1024)])
def test_batchnorm_residual_relu_backward_from_the_bit_map(rows, C):
    """ssl4gie_bn_fwd_partials_bits / ssl4gie_bn_bwd_bits (the ReLU mask of bn3 as one bit per element) against
    ssl4gie_bn_fwd_partials / ssl4gie_bn_bwd reading the ReLU output: y, dx, dres, dgamma, dbeta bit-identical"""
    from ssl4gie_amd import ops
    g = G(rows + C + 1)
    x = (torch.randn(rows, C, generator=g) * 2 + 1).to(BF).to(DEV)
    res = torch.randn(rows, C, generator=g).to(BF).to(DEV)
    gamma = torch.randn(C, generator=g).to(DEV)
    beta = (torch.randn(C, generator=g) * 0.3).to(DEV)
    x2 = x.float()
    xp = torch.cat([x2, x2.new_zeros((-rows) % 128, C)]).view(-1, 128, C)
    st = torch.stack([xp.sum(1), (xp * xp).sum(1)], 1).contiguous()
    rm0, rv0 = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    rm1, rv1 = rm0.clone(), rv0.clone()
    y0, mean0, rstd0 = ops.bn_fwd(x, gamma, beta, res, rm0, rv0, 0.1, 1e-5, True, True, partials=st)
    y1, bits, mean1, rstd1 = ops.bn_fwd_bits(x, gamma, beta, res, rm1, rv1, 0.1, 1e-5, st)
    assert torch.equal(y0, y1) and torch.equal(mean0, mean1) and torch.equal(rstd0, rstd1) and torch.equal(rm0, rm1)
    ref_bits = (y0.float() > 0).view(-1, 8).to(torch.int32)
    ref_bits = (ref_bits << torch.arange(8, device=DEV, dtype=torch.int32)).sum(1).to(torch.uint8)
    assert torch.equal(bits, ref_bits)
    dy = torch.randn(rows, C, generator=g).to(BF).to(DEV)
    dg0, db0 = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    dg1, db1 = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    dx0, dr0 = ops.bn_bwd(dy, y0, x, gamma, mean0, rstd0, True, True, dg0, db0, False)
    dx1, dr1 = ops.bn_bwd_bits(dy, bits, x, gamma, mean0, rstd0, dg1, db1, False)
    assert torch.equal(dx0, dx1) and torch.equal(dr0, dr1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)


def test_conv_operand_images_in_one_launch_equal_the_single_packs():
    """ssl4gie_conv3x3_weight_pack_batch against ssl4gie_conv3x3_weight_pack: all three modes, padded and unpadded
    rows, 70 weights (two kernel-argument batches); and a ResNet-50 step after a fused optimizer step gives the
    same output with the batched refresh as with the lazy single packs"""
    from ssl4gie_amd import dpt_engine, ops
    g = G(55)
    ws, modes, lds = [], [], []
    for i in range(70):
        co, ci = [(64, 64), (128, 32), (96, 256), (32, 128), (512, 64)][i % 5]
        ws.append(torch.randn(co, ci, 3, 3, generator=g).to(DEV))
        modes.append(i % 3)
        need = 9 * (co if i % 3 == 1 else ci)
        lds.append(None if i % 2 else need + 64)
    outs = ops.conv3x3_weight_pack_batch(ws, BF, modes, lds)
    for w, m, ld, o in zip(ws, modes, lds, outs):
        assert torch.equal(o, ops.conv3x3_weight_pack(w, BF, m, ld))
    from ssl4gie_amd.optim import ArenaAdamW
    res = []
    for batched in (True, False):
        dpt_engine._BATCH_PACK = batched
        m, _ = _resnet_pair(5)
        m.to(DEV).set_precision("bf16")
        m.train()
        opt = ArenaAdamW(m, list(m.parameters()), lr=1e-3)
        imgs = torch.randn(4, 3, 64, 64, generator=G(56)).to(DEV)
        for _ in range(2):
            opt.zero_grad()
            m(imgs).square().mean().backward()
            opt.step()
        with torch.no_grad():
            res.append(m(imgs).clone())
    dpt_engine._BATCH_PACK = True
    assert torch.equal(res[0], res[1])


def test_maxpool_avgpool_subsample():
    from ssl4gie_amd.resnet_engine import AvgPoolFn, MaxPoolFn, Subsample2Fn
    x = torch.relu(torch.randn(2, 16, 12, 10, generator=G(6)))  # ReLU output: many tied zeros
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 3, 2, 1)
    dy = torch.randn(yr.shape, generator=G(7))
    yr.backward(dy)
    xd = nhwc(x).to(DEV).requires_grad_(True)
    y = MaxPoolFn.apply(xd)
    y.backward(nhwc(dy).to(DEV))
    assert torch.equal(nchw(y.detach().cpu()), yr.detach())
    assert rel_err(nchw(xd.grad.cpu()), xr.grad) < 1e-6
    xr.grad = None
    pr = xr.mean((2, 3))
    dp = torch.randn(pr.shape, generator=G(8))
    pr.backward(dp)
    xd2 = nhwc(x).to(DEV).requires_grad_(True)
    p = AvgPoolFn.apply(xd2)
    p.backward(dp.to(DEV))
    assert rel_err(p.detach().cpu(), pr) < 1e-6
    assert rel_err(nchw(xd2.grad.cpu()), xr.grad) < 1e-6
    xd3 = nhwc(x).to(DEV).requires_grad_(True)
    s = Subsample2Fn.apply(xd3)
    assert torch.equal(nchw(s.detach().cpu()), x[:, :, ::2, ::2])
    s.backward(torch.ones_like(s))
    ref = torch.zeros_like(x)
    ref[:, :, ::2, ::2] = 1
    assert torch.equal(nchw(xd3.grad.cpu()), ref)


def _resnet_pair(seed, zero_init=False):
    from ssl4gie_amd.Models import models
    torch.manual_seed(seed)
    m = models.ResNet_from_Any(None, False, None, False, None)
    if zero_init:
        pass
    # non-trivial affine parameters so that dgamma / dbeta paths are exercised
    g = G(seed + 1)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.copy_(1 + 0.1 * torch.randn(mod.weight.shape, generator=g))
                mod.bias.copy_(0.1 * torch.randn(mod.bias.shape, generator=g))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    return m, sd


def test_resnet50_schema():
    m, sd = _resnet_pair(0)
    assert "fc.weight" not in sd and "conv1.weight" in sd
    assert sd["layer1.0.downsample.0.weight"].shape == (256, 64, 1, 1)
    assert sd["layer4.2.conv3.weight"].shape == (2048, 512, 1, 1)
    assert sd["layer2.0.conv2.weight"].shape == (128, 128, 3, 3)
    n_params = sum(p.numel() for p in m.parameters())
    assert n_params == 23508032  # torchvision resnet50 without fc
    assert sum(1 for k in sd if k.endswith("running_mean")) == 53


def _l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm())


def _oracle_run(sd, imgs, w, dt):
    from oracle import resnet_ref
    sdo = {k: (v.clone().to(dt).requires_grad_(True) if (v.is_floating_point() and "running" not in k)
               else v.clone()) for k, v in sd.items()}
    out = resnet_ref.resnet50_pooled(sdo, imgs.to(dt))
    (out * w.to(dt)).sum().backward()
    return out.detach(), {k: v.grad for k, v in sdo.items() if v.requires_grad and v.grad is not None}


NAMES = ("layer4.2.bn3.bias", "layer4.2.conv3.weight", "layer4.0.conv1.weight", "layer3.0.conv2.weight",
         "layer2.0.downsample.0.weight", "layer1.0.conv2.weight", "bn1.weight", "conv1.weight")


def test_resnet50_trunk_fp32_as_accurate_as_torch_fp32():
    """Random-init ResNet50 in training mode is ill-conditioned for a 1e-3 gradient check: a handful
    of ReLU masks flip between any two fp32 evaluations and BatchNorm's batch statistics spread that
    over every element (torch's own fp32 and fp64 CPU results differ by ~2 % in relative L2 on the
    early-layer gradients).  So the engine's fp32 path is held to the bar torch fp32 itself meets:
    its distance to the fp64 oracle may not exceed 2.5x torch-fp32's distance to the fp64 oracle;
    the forward (no discontinuity amplification) is held to 1e-3."""
    m, sd = _resnet_pair(1)
    m.to(DEV).set_precision("fp32")
    imgs = torch.randn(8, 3, 128, 128, generator=G(9))
    w = torch.randn(2048, generator=G(10))
    out = m(imgs.to(DEV))
    (out * w.to(DEV)).sum().backward()
    o64, g64 = _oracle_run(sd, imgs, w, torch.float64)
    o32, g32 = _oracle_run(sd, imgs, w, torch.float32)
    assert out.shape == (8, 2048)
    assert rel_err(out, o64) < 1e-3
    grads = dict(m.named_parameters())
    for name in NAMES:
        ours, torch32 = _l2(grads[name].grad, g64[name]), _l2(g32[name], g64[name])
        assert ours <= 2.5 * torch32 + 1e-4, (name, ours, torch32)


def test_resnet50_trunk_bf16_tracks_oracle():
    """bf16 activations: ~6 roundings per bottleneck compound to ~1.3x per block at random init (the
    reference's fp16 autocast path has the same dtype flow with a 3-bit wider mantissa); judged on
    relative L2 of the pooled features and of the last block's gradients."""
    m, sd = _resnet_pair(1)
    m.to(DEV).set_precision("bf16")
    imgs = torch.randn(8, 3, 128, 128, generator=G(9))
    w = torch.randn(2048, generator=G(10))
    out = m(imgs.to(DEV))
    (out * w.to(DEV)).sum().backward()
    o64, g64 = _oracle_run(sd, imgs, w, torch.float64)
    assert _l2(out, o64) < 0.3
    assert _l2(dict(m.named_parameters())["layer4.2.bn3.bias"].grad, g64["layer4.2.bn3.bias"]) < 0.15
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


def test_resnet_dense_decoder_vs_oracle():
    """ResNet_from_Any(dense="depth") (decoder_levels + output_conv on the four
    stage maps) against the fp64 oracle restatement: output, SSI loss, decoder gradients"""
    from oracle import dpt_ref, resnet_ref
    from ssl4gie_amd.Models.models import ResNet_from_Any
    from ssl4gie_amd.losses import ScaleAndShiftInvariantLoss
    torch.manual_seed(0)
    m = ResNet_from_Any(None, False, 1, False, "depth")
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for name, p in m.named_parameters():  # non-trivial BatchNorm affine and biases everywhere
            if p.dim() == 1:
                p.copy_((1 + 0.1 * torch.randn(p.shape, generator=g)) if name.endswith("weight")
                        else 0.1 * torch.randn(p.shape, generator=g))
    keys = sorted(m.state_dict().keys())
    assert "decoder_levels.0.blocks.0.identity.0.weight" in keys and "output_conv.5.bias" in keys
    assert not any(k.startswith("fc.") for k in keys)
    sd = {k: (v.detach().double().requires_grad_(True) if v.is_floating_point() and "running" not in k
              else v.detach().clone()) for k, v in m.state_dict().items()}
    m.to(DEV).set_precision("fp32")
    imgs = torch.randn(4, 3, 128, 128, generator=g)
    target = torch.rand(4, 1, 128, 128, generator=g)
    out = m(imgs.to(DEV))
    loss = ScaleAndShiftInvariantLoss(alpha=0.1)(out, target.to(DEV))
    loss.backward()
    out_o = resnet_ref.resnet50_dense(sd, imgs.double())
    loss_o = dpt_ref.ssi_loss(out_o, target.double(), alpha=0.1)
    loss_o.backward()
    assert out.shape == (4, 1, 128, 128)
    assert rel_err(out.cpu(), out_o.float()) < 2e-3
    assert abs(float(loss.detach()) - float(loss_o)) < 2e-3 * abs(float(loss_o))
    pg = dict(m.named_parameters())
    for name in ("output_conv.5.weight", "output_conv.3.weight", "output_conv.1.bias",
                 "decoder_levels.2.blocks.2.process.6.weight", "decoder_levels.2.blocks.0.identity.0.weight",
                 "decoder_levels.1.chan_reduce.0.weight", "decoder_levels.0.blocks.1.process.3.weight"):
        # fp32 engine vs fp64 oracle through a random-init ResNet50 at batch 4 (ill-conditioned, see
        # the trunk test above): several % of L2 noise; a structural error would be of order 1
        assert rel_err(pg[name].grad.cpu(), sd[name].grad.float()) < 0.15, name


@pytest.mark.parametrize("T,K,N", [(1000, 64, 64), (12544, 256, 128), (300, 128, 264), (128, 64, 8)])
def test_gemm_colstats_match_the_stored_outputs(T, K, N):
    """ssl4gie_gemm_desc.colstats: per-128-row column sums / sums of squares of the bf16 outputs the
    GEMM stored, and BatchNorm from them == BatchNorm with its own statistics pass"""
    from ssl4gie_amd import ops
    g = torch.Generator().manual_seed(50)
    x = torch.randn(T, K, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(torch.bfloat16).to(DEV)
    y, st = ops.linear_fwd(x, w, None, colstats=True)
    y0 = ops.linear_fwd(x, w, None)
    assert torch.equal(y, y0) and st.shape == ((T + 127) // 128, 2, N)
    yf = y.float()
    pad = (-T) % 128
    yp = torch.cat([yf, yf.new_zeros(pad, N)]).view(-1, 128, N)
    assert rel_err(st[:, 0], yp.sum(1)) < 1e-5
    assert rel_err(st[:, 1], (yp * yp).sum(1)) < 1e-5
    gam = (1 + 0.1 * torch.randn(N, generator=g)).to(DEV)
    bet = (0.1 * torch.randn(N, generator=g)).to(DEV)
    rm0, rv0 = torch.zeros(N, device=DEV), torch.ones(N, device=DEV)
    rm1, rv1 = rm0.clone(), rv0.clone()
    a, ma, ra = ops.bn_fwd(y, gam, bet, None, rm0, rv0, 0.1, 1e-5, True, True)
    b, mb, rb = ops.bn_fwd(y, gam, bet, None, rm1, rv1, 0.1, 1e-5, True, True, partials=st)
    assert rel_err(mb, ma) < 1e-4 and rel_err(rb, ra) < 1e-4
    assert rel_err(b.float(), a.float()) < 5e-3 and rel_err(rm1, rm0) < 1e-4 and rel_err(rv1, rv0) < 1e-4
    m2, v2 = ops.bn_stats(y, partials=st)   # the SyncBatchNorm half
    m1, v1 = ops.bn_stats(y)
    assert rel_err(m2, m1) < 1e-4 and rel_err(v2, v1) < 1e-4


@pytest.mark.parametrize("T,K,N,aux,relu", [(1000, 64, 256, True, True), (12544, 256, 1024, True, True),
                                            (300, 128, 264, False, False), (12800, 64, 768, True, False),
                                            (256, 64, 8, False, True), (4099, 512, 2048, True, True)])
def test_gemm_statistics_only_and_affine_epilogue(T, K, N, aux, relu):
    """the BatchNorm-fused 1x1 convolution's two products: (1) colstats with C == NULL == the statistics of the
    storing product, bit for bit, and nothing written; (2) EPI_AFFINE_AUX_RELU == act(acc scale + shift (+ aux))
    on the fp32 accumulators, rounded once (ragged rows, 192-wide tiles at N = 768, N not a multiple of 64)"""
    from ssl4gie_amd import ops
    g = torch.Generator().manual_seed(51 + T)
    x = torch.randn(T, K, generator=g).to(BF).to(DEV)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to(BF).to(DEV)
    _, st = ops.linear_fwd(x, w, None, colstats=True)
    st2 = ops.linear_colstats_only(x, w)
    assert torch.equal(st, st2)
    scale = (1 + 0.3 * torch.randn(N, generator=g)).to(DEV)
    scale[0] = -scale[0]
    shift = (0.2 * torch.randn(N, generator=g)).to(DEV)
    a = torch.randn(T, N, generator=g).to(BF).to(DEV) if aux else None
    y = ops.linear_affine_fwd(x, w, scale, shift, a, relu)
    ref = (x.double() @ w.double().t()) * scale.double() + shift.double()
    if aux:
        ref = ref + a.double()
    if relu:
        ref = ref.clamp_min(0)
    assert y.dtype == BF and y.shape == (T, N)
    err = (y.double() - ref).abs()
    # one bf16 rounding of the exact value (2^-9 relative) + fp32 accumulation noise
    assert bool((err <= 2.0 ** -8 * ref.abs() + 1e-5).all()), float((err / (ref.abs() + 1e-3)).max())
    if relu:
        assert float(y.float().min()) >= 0.0


def test_resnet50_no_grad_forward_recomputes_the_wide_products():
    """under torch.no_grad() (MoCo's momentum encoder) conv3 / downsample + BatchNorm (+ identity, ReLU) run as a
    statistics-only product and a product with the normalisation in its epilogue: the pooled features track the
    storing path (same statistics, one rounding less), the running statistics and batch counts agree"""
    import copy
    from ssl4gie_amd.Models import resnet as R
    m, _ = _resnet_pair(3)
    m.to(DEV).set_precision("bf16")
    m.train()
    m2 = copy.deepcopy(m)
    imgs = torch.randn(8, 3, 128, 128, generator=G(12)).to(DEV)
    calls = {"n": 0}
    orig = R.ResNet50._c1_bn_nograd

    def counted(self, *a, **k):
        calls["n"] += 1
        return orig(self, *a, **k)
    R.ResNet50._c1_bn_nograd = counted
    try:
        with torch.no_grad():
            a = m(imgs)
    finally:
        R.ResNet50._c1_bn_nograd = orig
    assert calls["n"] == 20          # 16 conv3 + 4 downsample
    b = m2(imgs)                      # gradients enabled: the storing path
    e = _l2(a, b.detach())
    print(f"pooled features, recomputing vs storing path: relative L2 {e:.3e}")
    sa, sb = m.state_dict(), m2.state_dict()
    for k in sa:
        if k.endswith("running_mean") or k.endswith("running_var"):
            # the first block's inputs are identical on both paths, and so are the statistics (same partial sums)
            tol = 1e-6 if k.startswith(("layer1.0.downsample", "layer1.0.bn")) else 0.3
            assert rel_err(sa[k].float().cpu(), sb[k].float().cpu()) < tol, k
        if k.endswith("num_batches_tracked"):
            assert int(sa[k]) == int(sb[k]) == 1, k
    # 16 random-init bottlenecks of bf16 roundings apart at batch 8: ill-conditioned (the bf16-vs-fp64 test above
    # allows 0.3 for the same reason; measured 0.12) — the arithmetic itself is held to one bf16 rounding by
    # test_gemm_statistics_only_and_affine_epilogue and to 1e-6 on the first block's statistics above
    assert e < 0.3


# bf16 engine on the same well-conditioned blocks: measured worst errors over the three blocks (round 3,
# profiles/r03p_bottleneck_fp32_bf16.log: output 8.2e-3, input gradient 6.2e-3, parameter gradients 2.5e-2 of the
# largest gradient of their kind) and bars at 1.5x them; with no mask flips what remains is operand rounding
# (8 significant bits, ~6 roundings per bottleneck) — the loose 15-30 % bars of the random-init trunk tests
# above are conditioning, not arithmetic.  The fp32 engine's own numbers on these blocks: 9e-7 / 9e-7 / 1.2e-4.
BF16_BLOCK_MEASURED = {"out": 8.2e-3, "dx": 6.2e-3, "param": 2.5e-2}


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("layer,idx,hw", [("layer1", 1, 28), ("layer2", 0, 28), ("layer3", 0, 14)])
def test_bottleneck_well_conditioned_fp32_every_gradient_1e3(layer, idx, hw, precision):
    """One torchvision Bottleneck (v1.5: stride on the 3x3; `Models/models.py:63-69`) in training mode on an
    input where NO ReLU pre-activation lies within 1e-2 of zero: the BatchNorm shifts are +-6 / +-12 per
    channel (half the channels always on, half always off), so no mask can flip between two evaluations
    and nothing amplifies rounding — the fp32 engine must then meet the north_star's 1e-3 on the output
    and on EVERY gradient (input, the three / four convolutions, every BatchNorm weight and bias).  The
    random-init whole-trunk tests above keep the loose, ill-conditioned bars; this one pins the arithmetic."""
    from ssl4gie_amd.Models.resnet import ResNet50
    torch.manual_seed(3)
    net = ResNet50()
    blk = getattr(net, layer)[idx]
    g = G(11)
    with torch.no_grad():
        for name, mod in blk.named_modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                c = mod.weight.numel()
                sign = torch.where(torch.arange(c) % 2 == 0, 1.0, -1.0)
                big = 12.0 if name in ("bn3",) else (0.0 if name.startswith("downsample") else 6.0)
                mod.weight.copy_(0.8 + 0.4 * torch.rand(c, generator=g))
                mod.bias.copy_(big * sign + 0.1 * torch.randn(c, generator=g))
    net.to(DEV).set_precision(precision)
    bars = {"out": 1e-3, "dx": 1e-3, "param": 1e-3} if precision == "fp32" else \
        {k: 1.5 * v for k, v in BF16_BLOCK_MEASURED.items()}
    cin = blk.conv1.weight.shape[1]
    stride, has_down = blk.conv2.stride[0], blk.downsample is not None
    x = torch.randn(4, cin, hw, hw, generator=g)
    # fp64 reference of the same block, keeping the ReLU pre-activations
    sd = {k: v.detach().clone().cpu().double().requires_grad_(v.is_floating_point() and "running" not in k and "num_batches" not in k)
          for k, v in blk.state_dict().items()}
    xr = x.double().requires_grad_(True)
    bn = lambda p, t: F.batch_norm(t, None, None, sd[p + ".weight"], sd[p + ".bias"], True, 0.1, 1e-5)
    p1 = bn("bn1", F.conv2d(xr, sd["conv1.weight"]))
    p2 = bn("bn2", F.conv2d(F.relu(p1), sd["conv2.weight"], stride=stride, padding=1))
    o3 = bn("bn3", F.conv2d(F.relu(p2), sd["conv3.weight"]))
    idn = bn("downsample.1", F.conv2d(xr, sd["downsample.0.weight"], stride=stride)) if has_down else xr
    p3 = o3 + idn
    yr = F.relu(p3)
    for pre in (p1, p2, p3):
        assert float(pre.detach().abs().min()) > 1e-2, "the construction must keep every pre-activation away from 0"
        frac_on = float((pre.detach() > 0).double().mean())
        assert 0.3 < frac_on < 0.7   # both ReLU branches are exercised
    dy = torch.randn(yr.shape, generator=g).double()
    yr.backward(dy)
    net._prepare()
    dt = torch.float32 if precision == "fp32" else torch.bfloat16
    xd = nhwc(x).to(DEV).to(dt).requires_grad_(True)
    y = net._block(xd, blk)
    y.backward(nhwc(dy.float()).to(DEV).to(dt))
    e_out, e_dx = rel_err(nchw(y.detach().float().cpu()), yr.detach()), rel_err(nchw(xd.grad.float().cpu()), xr.grad)
    assert e_out < bars["out"], e_out
    assert e_dx < bars["dx"], e_dx
    # Some gradients are analytically ZERO here (a BatchNorm shift in front of an always-on ReLU, a linear
    # convolution and another BatchNorm is removed by that BatchNorm's mean subtraction: bn2.bias, half of
    # bn1.bias; the fp64 reference holds 1e-13 there): an error relative to the tensor's own maximum means
    # nothing for them, so the denominator is floored at 1e-2 of the largest gradient among the parameters
    # of the same kind (all BatchNorm biases, all BatchNorm weights, all convolution weights).
    kind = lambda k: "bnb" if k.endswith(".bias") else ("bnw" if sd[k].dim() == 1 else "conv")
    scale = {}
    for name, _ in blk.named_parameters():
        scale[kind(name)] = max(scale.get(kind(name), 0.0), float(sd[name].grad.abs().max()))
    n, worst = 0, 0.0
    for name, p in blk.named_parameters():
        assert p.grad is not None, name
        ref = sd[name].grad
        # bf16: the analytically-zero gradients are sums of ~1e3 terms rounded to 8 bits each; they are judged
        # against the largest gradient of their kind (floor 1.0), the fp32 engine against 1e-2 of it
        den = max(float(ref.abs().max()), (1e-2 if precision == "fp32" else 1.0) * scale[kind(name)])
        err = float((p.grad.double().cpu() - ref).abs().max()) / den
        worst = max(worst, err)
        if precision == "fp32":
            assert err < bars["param"], (name, err)
        n += 1
    assert n == (12 if has_down else 9)
    assert worst < bars["param"], worst
    print(f"bottleneck[{layer}.{idx} {precision}]: out {e_out:.2e} dx {e_dx:.2e} worst parameter gradient {worst:.2e}")
