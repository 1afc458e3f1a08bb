"""GPU tests at the PRODUCTION shapes of BASELINE.json configs[1] (MAE ViT-B, B = 256 per GPU): the
GEMM kernels on the exact (M, N, K) the step launches, checked exactly with small-integer operands
(every product and partial sum is exact in bf16 / fp32, so any tile-assignment, split-K, tail or
swizzle slip at these sizes is a bit error), and one full-size MAE step checked through the
size-independent properties of the path: 147 masked patches per row, ids_restore o
ids_shuffle = identity, the masked tokens' input never reaches the encoder, finite loss and
gradients, loss identical for the same noise.  The independent reference for the GEMMs is rocBLAS
fp64 (torch.matmul on the device) — not the engine's own generic kernel."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ssl4gie_amd import _lib
    _lib.load()


def ints(shape, seed, lo=-2, hi=3):
    g = torch.Generator("cpu").manual_seed(seed)
    return torch.randint(lo, hi, shape, generator=g).float()


# (M, N, K): decoder fc1 / fc2 / qkv / pred, encoder fc2 / proj / qkv — SURVEY Appendix C
NT_PROD = [(50432, 2048, 512), (50432, 512, 2048), (50432, 1536, 512), (50432, 768, 512),
           (12800, 768, 3072), (12800, 768, 768), (12800, 2304, 768)]


@pytest.mark.parametrize("M,N,K", NT_PROD)
def test_nt_gemm_production_shape_exact(M, N, K):
    from ssl4gie_amd import _lib, ops
    x = ints((M, K), 101).to(DEV)
    w = ints((N, K), 102).to(DEV)
    bias = ints((N,), 103).to(DEV)
    ref = x.double() @ w.double().t() + bias.double()
    xb, wb = x.to(BF), w.to(BF)
    y = ops.linear_fwd(xb, wb, bias, out_dtype=F32)
    assert torch.equal(y.double(), ref), f"bias/fp32: max diff {(y.double() - ref).abs().max()}"
    res = ints((M, N), 104, -8, 9).to(DEV)
    y = ops.linear_fwd(xb, wb, bias, out_dtype=F32, epilogue=_lib.EPI_BIAS_RESIDUAL, residual=res)
    assert torch.equal(y.double(), ref + res.double()), "residual epilogue"
    if abs(ref).max() < 256:  # representable in bf16: the bf16-output epilogue is exact too
        y = ops.linear_fwd(xb, wb, bias, out_dtype=BF)
        assert torch.equal(y.double(), ref), "bf16 output"
    # data-gradient orientation of the same weight: dX[M, K] = dY[M, N] W[N, K]
    dy = ints((M, N), 105, -1, 2).to(DEV)
    refx = dy.double() @ w.double()
    dx = ops.linear_bwd_data(dy.to(BF), wb, wb.t().contiguous())
    if abs(refx).max() < 256:
        assert torch.equal(dx.double(), refx), "data gradient"
    else:
        assert (dx.double() - refx).abs().max() <= refx.abs().max() * 2 ** -8


def test_nt_gemm_192_wide_tiles_ragged_exact():
    """The 256 x 192 tile variant (chosen per shape when it means fewer chip rounds: N = 768 at M = 12800) on a
    shape with a ragged last row tile AND a ragged last column tile (712 = 3 x 192 + 136: the third wave column
    of the last tile is partial, the fourth empty): every epilogue kind, exact integers."""
    from ssl4gie_amd import _lib, ops
    M, N, K = 12700, 712, 768
    x = ints((M, K), 201, -1, 2).to(DEV)
    w = ints((N, K), 202, -1, 2).to(DEV)
    bias = ints((N,), 203).to(DEV)
    ref = x.double() @ w.double().t() + bias.double()
    xb, wb = x.to(BF), w.to(BF)
    y = ops.linear_fwd(xb, wb, bias, out_dtype=F32)
    assert torch.equal(y.double(), ref), f"bias/fp32: max diff {(y.double() - ref).abs().max()}"
    res = ints((M, N), 204, -8, 9).to(DEV)
    y = ops.linear_fwd(xb, wb, bias, out_dtype=F32, epilogue=_lib.EPI_BIAS_RESIDUAL, residual=res)
    assert torch.equal(y.double(), ref + res.double()), "residual epilogue"
    assert abs(ref).max() < 256
    y = ops.linear_fwd(xb, wb, bias, out_dtype=BF)
    assert torch.equal(y.double(), ref), "bf16 output"
    y = ops.linear_fwd(xb, wb, None, out_dtype=BF)
    assert torch.equal(y.double(), ref - bias.double()), "bf16 output, no bias"
    # GELU pair: u exact is not representable after GELU; compare with the torch formulation at bf16 resolution
    d, g = ops.linear_fwd(xb, wb, bias, epilogue=_lib.EPI_BIAS_GELU_GRAD)
    u = ref.float()
    g_ref = torch.nn.functional.gelu(u)
    cdf = 0.5 * (1 + torch.erf(u / 2 ** 0.5))
    d_ref = cdf + u * torch.exp(-0.5 * u * u) / (2 * torch.pi) ** 0.5
    assert (g.float() - g_ref).abs().max() <= 2 ** -7 * g_ref.abs().max() + 1e-3
    assert (d.float() - d_ref).abs().max() <= 2 ** -7 * 1.2 + 1e-3
    # multiply-by-aux epilogue (the backward of fc2 into the saved GELU derivative)
    aux = ints((M, N), 205, -1, 2).to(DEV).to(BF)
    dsc = ops._desc(M, N, K, _lib.BF16, _lib.BF16)
    out = torch.empty(M, N, dtype=BF, device=DEV)
    dsc.A, dsc.sAm, dsc.sAk = xb.data_ptr(), K, 1
    dsc.B, dsc.sBk, dsc.sBn = wb.data_ptr(), 1, K
    dsc.C, dsc.ldc = out.data_ptr(), N
    dsc.epilogue, dsc.aux = _lib.EPI_MUL_AUX, aux.data_ptr()
    ops.gemm_raw(dsc, DEV)
    assert torch.equal(out.double(), (ref - bias.double()) * aux.double()), "mul-aux epilogue"


# (n_out, k_in, T): dW of decoder fc1 / fc2 / qkv / proj and encoder fc2 / qkv
TN_PROD = [(2048, 512, 50432), (512, 2048, 50432), (1536, 512, 50432), (512, 512, 50432),
           (768, 3072, 12800), (2304, 768, 12800)]


@pytest.mark.parametrize("No,Ki,T", TN_PROD)
def test_tn_gemm_production_shape_exact(No, Ki, T):
    from ssl4gie_amd import ops
    dy = ints((T, No), 111, -1, 2).to(DEV)
    x = ints((T, Ki), 112, -2, 3).to(DEV)
    ref = dy.double().t() @ x.double()  # |sum| <= 2 T < 2^24: exact in fp32 whatever the split
    db = torch.empty(No, device=DEV)
    dw = ops.linear_bwd_weight(dy.to(BF), x.to(BF), bias_out=db)
    assert torch.equal(dw.double(), ref), f"max diff {(dw.double() - ref).abs().max()}"
    assert torch.equal(db.double(), dy.double().sum(0))
    dw2 = ops.linear_bwd_weight(dy.to(BF), x.to(BF), out=dw.clone(), accumulate=True)
    assert torch.equal(dw2.double(), 2 * ref)


@pytest.mark.parametrize("T,na,ka,nb,kb", [(50432, 512, 2048, 2048, 512), (50432, 512, 512, 1536, 512),
                                            (12800, 768, 3072, 3072, 768), (12800, 768, 768, 2304, 768)])
def test_tn_pair_production_shape_exact(T, na, ka, nb, kb):
    """the block executor's paired weight-gradient launches (dW_fc2 + dW_fc1, dW_proj + dW_qkv)"""
    from ssl4gie_amd import ops
    dya, xa = ints((T, na), 121, -1, 2).to(DEV), ints((T, ka), 122).to(DEV)
    dyb, xb = ints((T, nb), 123, -1, 2).to(DEV), ints((T, kb), 124).to(DEV)
    ba, bb = torch.empty(na, device=DEV), torch.empty(nb, device=DEV)
    wa, wb = ops.linear_bwd_weight_pair(dya.to(BF), xa.to(BF), dyb.to(BF), xb.to(BF), ba, bb)
    assert torch.equal(wa.double(), dya.double().t() @ xa.double())
    assert torch.equal(wb.double(), dyb.double().t() @ xb.double())
    assert torch.equal(ba.double(), dya.double().sum(0)) and torch.equal(bb.double(), dyb.double().sum(0))


def test_mae_vitb_full_batch_properties():
    """BASELINE.json configs[1] at full size (B = 256, bf16 engine), one step"""
    from ssl4gie_amd.Models.mae import models_mae
    torch.manual_seed(0)
    m = models_mae.mae_vit_base_patch16(norm_pix_loss=True).to(DEV).set_precision("bf16")
    B = 256
    imgs = torch.randn(B, 3, 224, 224, generator=torch.Generator("cpu").manual_seed(1)).to(DEV)
    noise = torch.rand(B, 196, generator=torch.Generator("cpu").manual_seed(2)).to(DEV)
    loss, pred, mask = m(imgs, mask_ratio=0.75, noise=noise)
    assert pred.shape == (B, 196, 768) and mask.shape == (B, 196)
    assert torch.equal(mask.sum(1), torch.full((B,), 147.0, device=DEV))  # int(196 * 0.25) = 49 kept
    assert bool(((mask == 0) | (mask == 1)).all())
    # the mask is the reference's rule on this noise: kept = the 49 smallest noise values per row
    ids_shuffle = torch.argsort(noise, dim=1, stable=True)
    ids_restore = torch.argsort(ids_shuffle, dim=1, stable=True)
    ar = torch.arange(196, device=DEV).expand(B, -1)
    assert torch.equal(torch.gather(ids_shuffle, 1, ids_restore), ar)  # ids_restore o ids_shuffle = id
    expect = torch.ones(B, 196, device=DEV)
    expect.scatter_(1, ids_shuffle[:, :49], 0.0)
    assert torch.equal(mask, expect)
    assert torch.equal(m._ids_shuffle.long(), ids_shuffle)
    assert torch.isfinite(loss) and 0.5 < float(loss) < 5.0
    loss.backward()
    for n, p in m.named_parameters():
        if p.requires_grad:
            assert p.grad is not None and bool(torch.isfinite(p.grad).all()), n
    # masked patches never reach the encoder: changing their pixels changes no kept-token latent
    with torch.no_grad():
        lat1, mask1, _ = m.forward_encoder(imgs, 0.75, noise=noise)
        pix = m.patchify(imgs)
        pix = torch.where(mask[:, :, None] > 0, pix + 3.0, pix)
        lat2, mask2, _ = m.forward_encoder(m.unpatchify(pix), 0.75, noise=noise)
        assert torch.equal(mask1, mask2) and torch.equal(lat1, lat2)
        # and the loss is a deterministic function of (images, noise)
        loss2, _, _ = m(imgs, mask_ratio=0.75, noise=noise)
    assert float(loss2) == float(loss)
