"""GPU parity tests, one op at a time, through the C ABI (ssl4gie_amd.ops -> libssl4gie_hip.so)
against CPU references (torch fp64 / the oracle).  Integer-valued operands make the bf16 MFMA
paths *exactly* checkable (catches any fragment-layout / swizzle / transposition slip)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ssl4gie_amd import _lib
    _lib.load()  # must be present on a GPU box: no fallback


def G(seed):
    return torch.Generator("cpu").manual_seed(seed)


def ints(shape, seed, lo=-2, hi=3):
    return torch.randint(lo, hi, shape, generator=G(seed)).float()


# ------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("cols", [128, 192, 512, 768, 1024])
@pytest.mark.parametrize("dt", [F32, BF])
def test_layernorm_fwd_bwd(cols, dt):
    from ssl4gie_amd import ops
    rows = 397
    x = torch.randn(rows, cols, generator=G(1)) * 2 + 0.5
    g = 1 + 0.1 * torch.randn(cols, generator=G(2))
    b = 0.1 * torch.randn(cols, generator=G(3))
    dy = torch.randn(rows, cols, generator=G(4))
    dres = torch.randn(rows, cols, generator=G(5))
    if dt == BF:
        dy = dy.bfloat16().float()
    xr = x.double().requires_grad_(True)
    gr, br = g.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.layer_norm(xr, (cols,), gr, br, 1e-6)
    yr.backward(dy.double())
    y, mean, rstd = ops.layernorm_fwd(x.to(DEV), g.to(DEV), b.to(DEV), 1e-6, dt)
    tol = 1e-5 if dt == F32 else 1e-2
    assert rel_err(y.float(), yr.detach()) < tol
    assert rel_err(mean, x.double().mean(1)) < 1e-5
    dx, dx_lp, dg, db = ops.layernorm_bwd(dy.to(DEV).to(dt), x.to(DEV), g.to(DEV), mean, rstd,
                                          dres=dres.to(DEV), want_lp=True)
    assert rel_err(dx, xr.grad + dres.double()) < 2e-5
    assert rel_err(dx_lp.float(), xr.grad + dres.double()) < tol
    assert rel_err(dg, gr.grad) < 2e-5 and rel_err(db, br.grad) < 2e-5
    # accumulate flag
    dx2, _, dg2, db2 = ops.layernorm_bwd(dy.to(DEV).to(dt), x.to(DEV), g.to(DEV), mean, rstd,
                                         dgamma=dg.clone(), dbeta=db.clone(), accumulate=True)
    assert rel_err(dg2, 2 * gr.grad) < 2e-5 and rel_err(dx2, xr.grad) < 2e-5


def test_colsum():
    from ssl4gie_amd import ops
    x = torch.randn(1577, 2304, generator=G(9))
    for dt, tol in ((F32, 1e-5), (BF, 1e-5)):
        xs = x.to(dt)
        out = ops.colsum(xs.to(DEV))
        assert rel_err(out, xs.double().sum(0)) < tol
        odd = xs[:37, :6].contiguous()  # 6-class head: not a multiple of 4 columns
        assert rel_err(ops.colsum(odd.to(DEV)), odd.double().sum(0)) < tol
    # tall and narrow (the 32-channel map in front of the depth head): the several-rows-per-wave
    # kernel, exact with small integers; row counts around the block / unroll boundaries
    for rows, cols in ((100003, 32), (5, 8), (4096 * 37 + 1, 4), (70001, 128), (2049, 64)):
        xi = torch.randint(-3, 4, (rows, cols), generator=G(10)).float()
        for dt in (F32, BF):
            got = ops.colsum(xi.to(dt).to(DEV))
            assert torch.equal(got.cpu().double(), xi.double().sum(0)), (rows, cols, dt)
        acc = torch.ones(cols, device=DEV)
        ops.colsum(xi.to(BF).to(DEV), out=acc, accumulate=True)
        assert torch.equal(acc.cpu().double(), xi.double().sum(0) + 1)


# ------------------------------------------------------------------ generic f32 GEMM
def _gemm_generic(A, B, M, N, K, sa, sb, batch=(1, 1), sab=(0, 0), sbb=(0, 0), alpha=1.0):
    from ssl4gie_amd import _lib, ops
    d = ops._desc(M, N, K, _lib.F32, _lib.F32)
    d.batch1, d.batch2 = batch
    d.A, d.sAm, d.sAk, d.sAb1, d.sAb2 = A.data_ptr(), sa[0], sa[1], sab[0], sab[1]
    d.B, d.sBk, d.sBn, d.sBb1, d.sBb2 = B.data_ptr(), sb[0], sb[1], sbb[0], sbb[1]
    C = torch.empty(batch[0], batch[1], M, N, device=DEV)
    d.C, d.ldc, d.sCb1, d.sCb2 = C.data_ptr(), N, batch[1] * M * N, M * N
    d.alpha = alpha
    ops.gemm_raw(d, DEV)
    return C


@pytest.mark.parametrize("M,N,K", [(64, 64, 16), (70, 50, 33), (197, 197, 64), (5, 6, 768)])
def test_gemm_generic_all_layouts(M, N, K):
    a = torch.randn(M, K, generator=G(1))
    b = torch.randn(K, N, generator=G(2))
    ref = a.double() @ b.double()
    ad, bd = a.to(DEV), b.to(DEV)
    at, bt = a.t().contiguous().to(DEV), b.t().contiguous().to(DEV)
    for A, sa in ((ad, (K, 1)), (at, (1, M))):
        for B, sb in ((bd, (N, 1)), (bt, (1, K))):
            C = _gemm_generic(A, B, M, N, K, sa, sb, alpha=0.5)
            assert rel_err(C[0, 0], 0.5 * ref) < 1e-5


def test_gemm_generic_batched():
    B1, B2, M, N, K = 2, 3, 50, 50, 64
    a = torch.randn(B1, B2, M, K, generator=G(3))
    b = torch.randn(B1, B2, N, K, generator=G(4))
    C = _gemm_generic(a.to(DEV), b.to(DEV), M, N, K, (K, 1), (1, K), batch=(B1, B2),
                      sab=(B2 * M * K, M * K), sbb=(B2 * N * K, N * K))
    assert rel_err(C, a.double() @ b.double().transpose(-1, -2)) < 1e-5


# ------------------------------------------------------------------ bf16 NT GEMM
NT_SHAPES = [(128, 128, 64), (400, 768, 768), (1576, 1536, 512), (130, 132, 128), (12, 2048, 512),
             (1000, 512, 2048),
             # large enough for the 256x256 ping-pong kernel's heuristic (>= 128 tiles), with ragged
             # M / N edges and 1, 2 and several K-tiles
             (4096, 2048, 64), (4000, 2056, 128), (8192, 1024, 512), (3000, 3072, 768)]


@pytest.mark.parametrize("M,N,K", NT_SHAPES)
def test_gemm_bf16_nt_exact_integers(M, N, K):
    """asymmetric small-integer operands: every product/sum is exact in bf16/fp32."""
    from ssl4gie_amd import ops
    x = ints((M, K), 11)
    w = ints((N, K), 12)
    bias = ints((N,), 13)
    ref = x.double() @ w.double().t() + bias.double()
    y = ops.linear_fwd(x.to(DEV).to(BF), w.to(DEV).to(BF), bias.to(DEV), out_dtype=F32)
    assert torch.equal(y.cpu().double(), ref), f"max diff {(y.cpu().double() - ref).abs().max()}"


@pytest.mark.parametrize("M,N,K", [(400, 768, 768), (1576, 2048, 512), (4352, 2048, 512),
                                   (4100, 2040, 256)])
def test_gemm_bf16_nt_epilogues(M, N, K):
    from ssl4gie_amd import _lib, ops
    x = (torch.randn(M, K, generator=G(1)) * 0.5).to(BF)
    w = (torch.randn(N, K, generator=G(2)) * 0.05).to(BF)
    bias = 0.1 * torch.randn(N, generator=G(3))
    res = torch.randn(M, N, generator=G(4))
    acc = x.double() @ w.double().t() + bias.double()
    xd, wd, bd = x.to(DEV), w.to(DEV), bias.to(DEV)
    y = ops.linear_fwd(xd, wd, bd, out_dtype=BF)
    assert rel_err(y.float(), acc) < 8e-3
    y = ops.linear_fwd(xd, wd, bd, out_dtype=F32, epilogue=_lib.EPI_BIAS_RESIDUAL,
                       residual=res.to(DEV))
    assert rel_err(y, acc + res.double()) < 1e-5
    u, g = ops.linear_fwd(xd, wd, bd, out_dtype=BF, epilogue=_lib.EPI_BIAS_GELU)
    assert rel_err(u.float(), acc) < 8e-3
    assert rel_err(g.float(), F.gelu(acc)) < 8e-3
    # data-gradient with GELU' epilogue: dx = (dy @ W) * gelu'(aux)
    dy = (torch.randn(M, N, generator=G(5)) * 0.5).to(BF)
    aux = torch.randn(M, K, generator=G(6)).to(BF)
    wt = w.t().contiguous()
    t = aux.double().requires_grad_(True)
    F.gelu(t).backward(dy.double() @ w.double())
    dx = ops.linear_bwd_data(dy.to(DEV), wd, wt.to(DEV), dgelu_aux=aux.to(DEV))
    assert rel_err(dx.float(), t.grad) < 1e-2
    # the executor's pair: forward saves gelu'(u) next to gelu(u), backward multiplies by it
    d1, g1 = ops.linear_fwd(xd, wd, bd, out_dtype=BF, epilogue=_lib.EPI_BIAS_GELU_GRAD)
    tt = acc.clone().requires_grad_(True)
    F.gelu(tt).sum().backward()
    assert rel_err(g1.float(), F.gelu(acc)) < 8e-3
    assert rel_err(d1.float(), tt.grad) < 8e-3
    dxm = ops.linear_bwd_data(dy.to(DEV), wd, wt.to(DEV), mul_aux=aux.to(DEV))
    assert rel_err(dxm.float(), (dy.double() @ w.double()) * aux.double()) < 1e-2
    dx2 = ops.linear_bwd_data(dy.to(DEV), wd, None)  # generic NN fallback agrees with NT fast path
    dx3 = ops.linear_bwd_data(dy.to(DEV), wd, wt.to(DEV))
    assert rel_err(dx2.float(), dx3.float()) < 8e-3


# ------------------------------------------------------------------ bf16 TN GEMM (weight grads)
TN_SHAPES = [(128, 128, 64), (768, 768, 400), (2304, 768, 1576), (512, 2048, 1000), (768, 768, 100),
             (136, 264, 12800), (8, 768, 70),
             # 256x256 ping-pong TN kernel (K % 64 == 0, K >= 1024): split-K, ragged M / N tiles
             (2304, 768, 1600), (520, 264, 2048), (512, 2048, 4096), (768, 768, 12800)]


@pytest.mark.parametrize("No,Ki,T", TN_SHAPES)
def test_gemm_bf16_tn_exact_integers(No, Ki, T):
    from ssl4gie_amd import ops
    dy = ints((T, No), 21, -1, 2)
    x = ints((T, Ki), 22, -2, 3)
    ref = dy.double().t() @ x.double()
    dw = ops.linear_bwd_weight(dy.to(DEV).to(BF), x.to(DEV).to(BF))
    assert torch.equal(dw.cpu().double(), ref), f"max diff {(dw.cpu().double() - ref).abs().max()}"
    dw2 = ops.linear_bwd_weight(dy.to(DEV).to(BF), x.to(DEV).to(BF), out=dw.clone(), accumulate=True)
    assert torch.equal(dw2.cpu().double(), 2 * ref)
    # bias gradient (column sums of dy) riding on the same product, overwrite and accumulate
    db = torch.full((No,), 7.0, device=DEV)
    dw3 = ops.linear_bwd_weight(dy.to(DEV).to(BF), x.to(DEV).to(BF), bias_out=db)
    assert torch.equal(dw3.cpu().double(), ref)
    assert torch.equal(db.cpu().double(), dy.double().sum(0))
    ops.linear_bwd_weight(dy.to(DEV).to(BF), x.to(DEV).to(BF), out=dw3, accumulate=True, bias_out=db)
    assert torch.equal(db.cpu().double(), 2 * dy.double().sum(0))


def test_gemm_f32_weight_grad_generic():
    from ssl4gie_amd import ops
    dy = torch.randn(333, 96, generator=G(1))
    x = torch.randn(333, 40, generator=G(2))
    dw = ops.linear_bwd_weight(dy.to(DEV), x.to(DEV))
    assert rel_err(dw, dy.double().t() @ x.double()) < 1e-5


# ------------------------------------------------------------------ attention
def _attn_ref(qkv, B, N, H, hd):
    q, k, v = qkv.reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4).unbind(0)
    s = (q @ k.transpose(-2, -1)) * hd ** -0.5
    lse = torch.logsumexp(s, dim=-1)
    o = (s.softmax(-1) @ v).transpose(1, 2).reshape(B, N, H * hd)
    return o, lse


@pytest.mark.parametrize("N,H,hd", [(50, 3, 64), (197, 2, 32), (17, 2, 64)])
def test_attention_f32_path(N, H, hd):
    from ssl4gie_amd import ops
    B = 2
    qkv = torch.randn(B, N, 3 * H * hd, generator=G(1))
    do = torch.randn(B, N, H * hd, generator=G(2))
    t = qkv.double().requires_grad_(True)
    o_ref, lse_ref = _attn_ref(t, B, N, H, hd)
    o_ref.backward(do.double())
    o, lse = ops.attn_fwd(qkv.to(DEV), B, N, H, hd)
    assert rel_err(o, o_ref.detach()) < 1e-5 and rel_err(lse, lse_ref.detach()) < 1e-5
    dqkv = ops.attn_bwd(qkv.to(DEV), o, do.to(DEV), lse, B, N, H, hd)
    assert rel_err(dqkv, t.grad) < 2e-5


# 1, 16, 40, 65, 129, 193, 197, 208: the last 16-key tile of the last pair is all padding (the kernels' HT variants);
# 209: one key in it
@pytest.mark.parametrize("N", [50, 197, 17, 64, 65, 128, 129, 224, 256, 300, 1, 16, 40, 193, 208, 209])
@pytest.mark.parametrize("hd", [64, 32])
def test_attention_bf16_fused(N, hd):
    from ssl4gie_amd import ops
    B, H = 2, 3
    qkv = (torch.randn(B, N, 3 * H * hd, generator=G(3)) * 1.5).to(BF)
    do = torch.randn(B, N, H * hd, generator=G(4)).to(BF)
    t = qkv.double().requires_grad_(True)
    o_ref, lse_ref = _attn_ref(t, B, N, H, hd)
    o, lse = ops.attn_fwd(qkv.to(DEV), B, N, H, hd)
    assert rel_err(o.float(), o_ref.detach()) < 1.5e-2
    assert rel_err(lse, lse_ref.detach()) < 2e-3
    # backward: feed the kernel's own bf16 O (what the engine does); reference uses exact O
    o_ref.backward(do.double())
    dqkv = ops.attn_bwd(qkv.to(DEV), o, do.to(DEV), lse, B, N, H, hd)
    D = H * hd
    ref = t.grad.reshape(B, N, 3, D)
    got = dqkv.float().cpu().reshape(B, N, 3, D)
    for i, name in enumerate("qkv"):
        assert rel_err(got[:, :, i], ref[:, :, i]) < 3e-2, f"d{name}"


@pytest.mark.parametrize("B,N,H", [(30, 197, 12), (43, 197, 12), (27, 256, 12), (25, 160, 12), (64, 224, 6)])
def test_attention_bf16_backward_persistent_prefetch_is_per_head_exact(B, N, H):
    """hd 64 above 144 tokens: with more heads than the chip holds workgroups the backward walks several heads per
    workgroup and fetches the next head while it finishes the current one (attn_bwd_pf_bf16_kernel).  A head's result
    must not depend on where in that walk it was computed: the whole batch at once == the same heads in launches of
    8 images (at most 96 heads: one head per workgroup, no prefetch), bit for bit; and the whole batch is right
    against the double-precision reference on a few images."""
    from ssl4gie_amd import ops
    hd = 64
    qkv = (torch.randn(B, N, 3 * H * hd, generator=G(11)) * 1.5).to(BF).to(DEV)
    do = torch.randn(B, N, H * hd, generator=G(12)).to(BF).to(DEV)
    o, lse = ops.attn_fwd(qkv, B, N, H, hd)
    full = ops.attn_bwd(qkv, o, do, lse, B, N, H, hd)
    for b0 in range(0, B, 8):
        b1 = min(B, b0 + 8)
        part = ops.attn_bwd(qkv[b0:b1].contiguous(), o[b0:b1].contiguous(), do[b0:b1].contiguous(),
                            lse[b0:b1].contiguous(), b1 - b0, N, H, hd)
        assert torch.equal(part, full[b0:b1]), f"images {b0}..{b1 - 1}"
    for b in (0, B // 2, B - 1):  # first, a middle and the last image: first / later / last head of a walk
        t = qkv[b:b + 1].double().cpu().requires_grad_(True)
        o_ref, _ = _attn_ref(t, 1, N, H, hd)
        o_ref.backward(do[b:b + 1].double().cpu())
        D = H * hd
        ref = t.grad.reshape(1, N, 3, D)
        got = full[b:b + 1].float().cpu().reshape(1, N, 3, D)
        for i, name in enumerate("qkv"):
            assert rel_err(got[:, :, i], ref[:, :, i]) < 3e-2, f"image {b} d{name}"


@pytest.mark.parametrize("B,N,H", [(2, 512, 3), (1, 1024, 2), (1, 4096, 1)])
def test_attention_bf16_long_sequences(B, N, H):
    """streaming kernels (N % 128 == 0, hd 64): the detection backbone's global attention"""
    from ssl4gie_amd import ops
    hd = 64
    qkv = (torch.randn(B, N, 3 * H * hd, generator=G(6)) * 1.5).to(BF)
    do = torch.randn(B, N, H * hd, generator=G(7)).to(BF)
    t = qkv.double().requires_grad_(True)
    o_ref, lse_ref = _attn_ref(t, B, N, H, hd)
    o, lse = ops.attn_fwd(qkv.to(DEV), B, N, H, hd)
    assert rel_err(o.float(), o_ref.detach()) < 1.5e-2
    assert rel_err(lse, lse_ref.detach()) < 2e-3
    o_ref.backward(do.double())
    dqkv = ops.attn_bwd(qkv.to(DEV), o, do.to(DEV), lse, B, N, H, hd)
    D = H * hd
    ref = t.grad.reshape(B, N, 3, D)
    got = dqkv.float().cpu().reshape(B, N, 3, D)
    for i, name in enumerate("qkv"):
        assert rel_err(got[:, :, i], ref[:, :, i]) < 3e-2, f"d{name}"


def test_attention_f32_long_sequence():
    from ssl4gie_amd import ops
    B, N, H, hd = 1, 384, 2, 64
    qkv = torch.randn(B, N, 3 * H * hd, generator=G(8))
    do = torch.randn(B, N, H * hd, generator=G(9))
    t = qkv.double().requires_grad_(True)
    o_ref, lse_ref = _attn_ref(t, B, N, H, hd)
    o_ref.backward(do.double())
    o, lse = ops.attn_fwd(qkv.to(DEV), B, N, H, hd)
    assert rel_err(o, o_ref.detach()) < 1e-5 and rel_err(lse, lse_ref.detach()) < 1e-5
    dqkv = ops.attn_bwd(qkv.to(DEV), o, do.to(DEV), lse, B, N, H, hd)
    assert rel_err(dqkv, t.grad) < 2e-5


def test_attention_bf16_one_hot_exact():
    """V = one-hot rows, scores forced to pick one key: O must equal the selected V row exactly."""
    from ssl4gie_amd import ops
    B, N, H, hd = 1, 197, 1, 64
    q = torch.zeros(B, N, hd)
    k = torch.zeros(B, N, hd)
    v = torch.zeros(B, N, hd)
    perm = torch.randperm(N, generator=G(5))
    for i in range(N):  # query i matches key perm[i] strongly on a private coordinate pair
        q[0, i, i % hd] = 8.0
        q[0, i, (i // hd + 7) % hd] += 8.0
    k[0, perm] = q[0]
    for j in range(N):
        v[0, j, j % hd] = float(1 + j // hd)
    qkv = torch.stack([q, k, v], dim=2).reshape(B, N, 3 * hd).to(BF)
    o, _ = ops.attn_fwd(qkv.to(DEV), B, N, H, hd)
    t = qkv.double()
    o_ref, _ = _attn_ref(t, B, N, H, hd)
    assert rel_err(o.float(), o_ref) < 1e-2


# ------------------------------------------------------------------ casts
def test_casts():
    from ssl4gie_amd import ops
    x = torch.randn(768, 3072, generator=G(1))
    assert torch.equal(ops.cast(x.to(DEV), BF).cpu(), x.to(BF))
    assert torch.equal(ops.cast_transpose(x.to(DEV), BF).cpu(), x.t().contiguous().to(BF))
    y = torch.randn(45, 70, generator=G(2))
    assert torch.equal(ops.cast_transpose(y.to(DEV), F32).cpu(), y.t().contiguous())
    a = torch.randn(197 * 4, generator=G(3))
    b = torch.randn(197 * 4, generator=G(4))
    o, olp = ops.add_cast(a.to(DEV), b.to(DEV), True, BF)
    assert torch.equal(o.cpu(), a + b) and torch.equal(olp.cpu(), (a + b).to(BF))


# ------------------------------------------------------------------ MAE glue (integer paths exact)
def test_mask_argsort_bit_exact_vs_reference_fixture():
    from ssl4gie_amd import ops
    g = load_golden("g1_masking.npz")
    s, r, m = ops.mask_argsort(torch.from_numpy(g["noise"]).to(DEV), 49)
    assert s.dtype == torch.int64
    assert np.array_equal(s.cpu().numpy(), g["ids_shuffle"])
    assert np.array_equal(r.cpu().numpy(), g["ids_restore"])
    assert np.array_equal(m.cpu().numpy(), g["mask"])
    s, r, m = ops.mask_argsort(torch.from_numpy(g["tie_noise"]).to(DEV), 49)
    assert np.array_equal(s.cpu().numpy(), g["tie_ids_shuffle"])
    assert np.array_equal(r.cpu().numpy(), g["tie_ids_restore"])
    assert np.array_equal(m.cpu().numpy(), g["tie_mask"])
    s, r, m = ops.mask_argsort(torch.zeros(0, 196, device=DEV), 49)  # empty batch
    assert s.shape == (0, 196)


def test_patch_gather_exact():
    from ssl4gie_amd import ops
    g = load_golden("g2_patchify.npz")
    imgs = torch.from_numpy(g["imgs"])
    p = ops.patch_gather(imgs.to(DEV), 16, order=1)
    assert np.array_equal(p.cpu().numpy().reshape(g["patches"].shape), g["patches"])
    cols = ops.patch_gather(imgs.to(DEV), 16, order=0)
    ref = F.unfold(imgs, kernel_size=16, stride=16).transpose(1, 2).reshape(-1, 768)
    assert torch.equal(cols.cpu(), ref)
    ids = torch.tensor([[3, 0, 15, 7], [1, 2, 14, 9]])
    sel = ops.patch_gather(imgs.to(DEV), 16, ids=ids.to(DEV), nsel=3, out_dtype=BF)
    ref3 = ref.reshape(2, 16, 768)[torch.arange(2)[:, None], ids[:, :3]].reshape(-1, 768).to(BF)
    assert torch.equal(sel.cpu(), ref3)


def test_token_assembly_roundtrip():
    from ssl4gie_amd import ops
    B, L, keep, D = 3, 16, 4, 128
    noise = torch.rand(B, L, generator=G(1))
    s, r, m = ops.mask_argsort(noise.to(DEV), keep)
    y = torch.randn(B * keep, D, generator=G(2))
    cls = torch.randn(D, generator=G(3))
    pos = torch.randn(L + 1, D, generator=G(4))
    x = ops.tokens_assemble(y.to(DEV), cls.to(DEV), pos.to(DEV), B, keep, ids=s)
    sc = s.cpu()
    ref = torch.empty(B, keep + 1, D)
    ref[:, 0] = cls + pos[0]
    for b in range(B):
        for j in range(keep):
            ref[b, 1 + j] = y[b * keep + j] + pos[1 + sc[b, j]]
    assert torch.equal(x.cpu(), ref)
    dx = torch.randn(B, keep + 1, D, generator=G(5))
    dcls = torch.zeros(D, device=DEV)
    dy = ops.tokens_assemble_bwd(dx.to(DEV), F32, dcls_out=dcls)
    assert torch.equal(dy.cpu(), dx[:, 1:].reshape(-1, D))
    assert rel_err(dcls, dx[:, 0].sum(0)) < 1e-6
    # decoder side
    yd = torch.randn(B, keep + 1, D, generator=G(6))
    mt = torch.randn(D, generator=G(7))
    xd = ops.decoder_assemble(yd.to(DEV), mt.to(DEV), pos.to(DEV), r, keep)
    rc = r.cpu()
    refd = torch.empty(B, L + 1, D)
    refd[:, 0] = yd[:, 0] + pos[0]
    for b in range(B):
        for i in range(L):
            src = yd[b, 1 + rc[b, i]] if rc[b, i] < keep else mt
            refd[b, 1 + i] = src + pos[1 + i]
    assert torch.equal(xd.cpu(), refd)
    dxd = torch.randn(B, L + 1, D, generator=G(8))
    dmt = torch.zeros(D, device=DEV)
    dyd = ops.decoder_assemble_bwd(dxd.to(DEV), s, keep, F32, dmask_out=dmt)
    t_y = yd.clone().requires_grad_(True)
    t_m = mt.clone().requires_grad_(True)
    full = torch.cat([t_y[:, 1:], t_m.expand(B, L - keep, D)], dim=1)
    full = torch.gather(full, 1, rc[:, :, None].expand(-1, -1, D))
    torch.cat([t_y[:, :1], full], 1).backward(dxd)
    assert rel_err(dyd, t_y.grad) < 1e-6 and rel_err(dmt, t_m.grad) < 1e-5


@pytest.mark.parametrize("norm_pix", [False, True])
def test_mae_loss_and_gradient(norm_pix):
    from oracle import mae_ref
    from ssl4gie_amd import ops
    B, p = 3, 16
    cfg = mae_ref.MAEConfig(img_size=64, norm_pix_loss=norm_pix)
    imgs = torch.randn(B, 3, 64, 64, generator=G(1))
    pred = torch.randn(B, 17, 768, generator=G(2))
    mask = (torch.rand(B, 16, generator=G(3)) > 0.3).float()
    t = pred.clone().requires_grad_(True)
    loss = mae_ref.mae_loss(cfg, imgs, t[:, 1:], mask)
    loss.backward()
    pp = ops.mae_loss_fwd(pred.to(DEV), imgs.to(DEV), mask.to(DEV), p, norm_pix)
    assert abs(float(pp.sum() / mask.sum()) - float(loss)) < 1e-5 * abs(float(loss))
    gpp = torch.full((B, 16), 1.0 / float(mask.sum()), device=DEV)
    dp = ops.mae_loss_bwd(pred.to(DEV), imgs.to(DEV), mask.to(DEV), p, norm_pix, gpp)
    assert rel_err(dp, t.grad) < 1e-5
    # without the cls row
    pp2 = ops.mae_loss_fwd(pred[:, 1:].contiguous().to(DEV), imgs.to(DEV), mask.to(DEV), p, norm_pix)
    assert torch.equal(pp2, pp)


# ------------------------------------------------------------------ block executor
@pytest.mark.parametrize("dt,N,D,H", [(F32, 50, 192, 3), (F32, 197, 128, 4), (BF, 50, 768, 12),
                                      (BF, 197, 512, 16)])
def test_block_stack_fwd_bwd(dt, N, D, H):
    from oracle import mae_ref, synth
    from ssl4gie_amd import engine
    from ssl4gie_amd.Models.vit_layers import Block
    import torch.nn as nn
    from functools import partial
    B, depth = 2, 2
    blocks = nn.ModuleList([Block(D, H, 4.0, norm_layer=partial(nn.LayerNorm, eps=1e-6))
                            for _ in range(depth)])
    sd = {}
    for i in range(depth):
        for k, shp in synth._block_shapes(D, 4 * D).items():
            sd[f"{i}.{k}"] = synth.synth_tensor(f"blocks.{i}.{k}", shp, 5)
    blocks.load_state_dict(sd)
    blocks.to(DEV)
    x = torch.randn(B, N, D, generator=G(1))
    dy = torch.randn(B, N, D, generator=G(2))
    dtap = torch.randn(B, N, D, generator=G(3))
    # oracle
    sdo = {k: v.double().requires_grad_(True) for k, v in sd.items()}
    xo = x.double().requires_grad_(True)
    h0 = mae_ref.block_fwd(sdo, "0.", xo, H, 1e-6)
    h1 = mae_ref.block_fwd(sdo, "1.", h0, H, 1e-6)
    ((h1 * dy.double()).sum() + (h0 * dtap.double()).sum()).backward()
    # engine
    xg = x.to(DEV).requires_grad_(True)
    out, taps = engine.run_blocks(blocks, xg, H, 1e-6, dt, engine.GradSink(None), taps=(0,))
    ((out * dy.to(DEV)).sum() + (taps[0] * dtap.to(DEV)).sum()).backward()
    tol_f, tol_g = (2e-5, 1e-4) if dt == F32 else (2e-2, 5e-2)
    assert rel_err(out, h1.detach()) < tol_f and rel_err(taps[0], h0.detach()) < tol_f
    assert rel_err(xg.grad, xo.grad) < tol_g
    for name, p in blocks.named_parameters():
        assert p.grad is not None, name
        assert rel_err(p.grad, sdo[name].grad) < tol_g, name


def test_normalize_u8_matches_totensor_normalize():
    from ssl4gie_amd import ops
    img = torch.randint(0, 256, (3, 30, 28, 3), generator=G(11), dtype=torch.uint8)
    out = ops.normalize_u8(img.to(DEV)).cpu()
    mean = torch.tensor(ops.IMAGENET_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(ops.IMAGENET_STD).view(1, 3, 1, 1)
    ref = (img.permute(0, 3, 1, 2).float() / 255.0 - mean) / std
    assert out.shape == ref.shape and rel_err(out, ref) < 1e-6


@pytest.mark.parametrize("T,na,ka,nb,kb", [(12800, 768, 3072, 3072, 768), (4096, 512, 512, 1536, 512),
                                            (1024, 264, 136, 72, 520), (256, 64, 64, 64, 64)])
def test_weight_gradient_pair_matches_two_products(T, na, ka, nb, kb):
    """ssl4gie_gemm_tn_pair: two TN products in one split-K launch == the two products alone"""
    from ssl4gie_amd import ops
    dya = torch.randn(T, na, generator=G(30)).to(BF).to(DEV)
    xa = torch.randn(T, ka, generator=G(31)).to(BF).to(DEV)
    dyb = torch.randn(T, nb, generator=G(32)).to(BF).to(DEV)
    xb = torch.randn(T, kb, generator=G(33)).to(BF).to(DEV)
    ba = torch.empty(na, device=DEV)
    bb = torch.empty(nb, device=DEV)
    wa, wb = ops.linear_bwd_weight_pair(dya, xa, dyb, xb, ba, bb)
    assert rel_err(wa, dya.double().t() @ xa.double()) < 1e-5
    assert rel_err(wb, dyb.double().t() @ xb.double()) < 1e-5
    assert rel_err(ba, dya.double().sum(0)) < 1e-5 and rel_err(bb, dyb.double().sum(0)) < 1e-5
    wa2, wb2 = ops.linear_bwd_weight_pair(dya, xa, dyb, xb)  # without the fused bias gradients
    assert rel_err(wa2, wa) < 1e-6 and rel_err(wb2, wb) < 1e-6


@pytest.mark.parametrize("T,shapes", [
    (12800, [(768, 3072), (3072, 768), (768, 768), (2304, 768)] * 2),   # two encoder blocks: 216 tiles, no split-K
    (4096, [(512, 2048), (2048, 512), (512, 512), (1536, 512)]),        # one small block: split-K slabs
    (1024, [(264, 136), (72, 520), (64, 64)]),                          # ragged tiles
    (200, [(64, 64), (128, 40)])])                                      # K not a multiple of 64: per-product fallback
def test_weight_gradient_group_matches_single_products(T, shapes):
    """ssl4gie_gemm_tn_group: n TN products in one launch == each product alone (exact with integer
    operands), bias gradients riding along, overwrite and accumulate"""
    from ssl4gie_amd import ops
    pairs, refs = [], []
    for j, (n_out, k_in) in enumerate(shapes):
        dy = ints((T, n_out), 300 + j, -1, 2)
        x = ints((T, k_in), 400 + j, -2, 3)
        refs.append((dy.double().t() @ x.double(), dy.double().sum(0)))
        pairs.append((dy.to(BF).to(DEV), x.to(BF).to(DEV), torch.empty(n_out, device=DEV) if j % 2 == 0 else None))
    outs = ops.linear_bwd_weight_group(pairs)
    for (dw, (rw, rb), (_, _, b)) in zip(outs, refs, pairs):
        assert torch.equal(dw.cpu().double(), rw)
        if b is not None:
            assert torch.equal(b.cpu().double(), rb)
    outs2 = ops.linear_bwd_weight_group(pairs, accumulate_into=[o.clone() for o in outs])
    for (dw, (rw, rb), (_, _, b)) in zip(outs2, refs, pairs):
        assert torch.equal(dw.cpu().double(), 2 * rw)
        if b is not None:
            assert torch.equal(b.cpu().double(), 2 * rb)


@pytest.mark.parametrize("group", [1, 2, 3])
def test_block_stack_deferred_weight_gradients_match_inline(group, monkeypatch):
    """BlockStackFn with the weight gradients of `group` blocks collected into one launch on the side
    stream (SSL4GIE_WGRAD_GROUP) == the per-block paired launches: same outputs, same gradients
    (bf16 engine; the products are the same sums in a different split-K order)"""
    import torch.nn as nn
    from ssl4gie_amd import engine
    from ssl4gie_amd.Models.vit_layers import Block
    B, N, D, H, depth = 8, 64, 128, 4, 5
    torch.manual_seed(3)
    blocks = nn.ModuleList([Block(D, H, 4.0, qkv_bias=True, norm_layer=lambda d: nn.LayerNorm(d, eps=1e-6))
                            for _ in range(depth)]).to(DEV)
    x = torch.randn(B, N, D, generator=G(5)).to(DEV)
    wgt = torch.randn(B, N, D, generator=G(6)).to(DEV)
    res = {}
    for g in (0, group):
        monkeypatch.setenv("SSL4GIE_WGRAD_GROUP", str(g))
        for p in blocks.parameters():
            p.grad = None
        xi = x.clone().requires_grad_(True)
        y, taps = engine.run_blocks(blocks, xi, H, 1e-6, BF, engine.GradSink(None), taps=(1,))
        ((y * wgt).sum() + taps[0].sum()).backward()
        torch.cuda.synchronize()
        res[g] = (y.detach().clone(), xi.grad.clone(), [p.grad.clone() for p in blocks.parameters()])
    assert torch.equal(res[0][0], res[group][0])
    assert rel_err(res[group][1], res[0][1]) < 1e-6
    for a, b in zip(res[group][2], res[0][2]):
        assert rel_err(a, b) < 2e-5


def test_compute_cus_setting_changes_grids_not_results():
    from ssl4gie_amd import _lib, ops
    L = _lib.load()
    x = torch.randn(12800, 768, generator=G(40)).to(BF).to(DEV)
    w = torch.randn(2304, 768, generator=G(41)).to(BF).to(DEV)
    dy = torch.randn(12800, 2304, generator=G(42)).to(BF).to(DEV)
    try:
        outs = []
        for cus in (256, 200, 64):
            assert L.ssl4gie_set_compute_cus(cus) == 0
            outs.append((ops.linear_fwd(x, w).float(), ops.linear_bwd_weight(dy, x)))
        assert L.ssl4gie_set_compute_cus(4) == 1000 and L.ssl4gie_set_compute_cus(300) == 1000
    finally:
        L.ssl4gie_set_compute_cus(240)
    for y, dw in outs[1:]:
        assert torch.equal(y, outs[0][0])          # NT: tile assignment changes, arithmetic does not
        assert rel_err(dw, outs[0][1]) < 2e-5      # TN: split-K boundaries move (fp32 summation order)
