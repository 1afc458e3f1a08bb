"""Host-side wrappers of the single-op C-ABI entry points (raw device pointers + current stream).

Every wrapper validates shapes / dtypes / contiguity on the host before the launch — a kernel is
never started on operands whose layout it does not assume.  Tensors are only containers for
device memory here; no torch arithmetic happens in this module.
"""
from __future__ import annotations

import ctypes as C

import os

import torch

from . import _lib
from ._lib import BF16, F32, GemmDesc

_TORCH2CODE = {torch.float32: F32, torch.bfloat16: BF16}
_CODE2TORCH = {F32: torch.float32, BF16: torch.bfloat16}


def code(dtype) -> int:
    try:
        return _TORCH2CODE[dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype {dtype}; the HIP path computes in fp32 or bf16")


def torch_dtype(c: int):
    return _CODE2TORCH[c]


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


def _dev(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("ssl4gie_amd ops need tensors on the HIP device (no CPU fallback)")
        if not t.is_contiguous():
            raise RuntimeError("ssl4gie_amd ops need contiguous tensors")


def _f32(*ts):
    for t in ts:
        if t is not None and t.dtype != torch.float32:
            raise TypeError(f"expected float32, got {t.dtype}")


# ------------------------------------------------------------------ LayerNorm
def layernorm_fwd(x, gamma, beta, eps, out_dtype, save_stats=True):
    _dev(x, gamma, beta)
    _f32(x, gamma, beta)
    cols = x.shape[-1]
    rows = x.numel() // cols
    assert gamma.numel() == cols and beta.numel() == cols and cols % 4 == 0
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if save_stats else None
    L = _lib.load()
    _lib.check(L.ssl4gie_layernorm_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(y), code(out_dtype),
                                       ptr(mean), ptr(rstd), rows, cols, float(eps), stream()),
               "layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, dres=None, want_lp=False, dgamma=None, dbeta=None,
                  accumulate=False):
    _dev(dy, x, gamma, mean, rstd, dres, dgamma, dbeta)
    _f32(x, gamma, mean, rstd, dres, dgamma, dbeta)
    cols = x.shape[-1]
    rows = x.numel() // cols
    assert dy.shape == x.shape and mean.numel() == rows and rstd.numel() == rows
    assert dres is None or dres.shape == x.shape
    L = _lib.load()
    dx = torch.empty_like(x)
    dx_lp = torch.empty(x.shape, dtype=dy.dtype, device=x.device) if want_lp else None
    if dgamma is None:
        dgamma = torch.empty(cols, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(cols, dtype=torch.float32, device=x.device)
    ws = torch.empty(L.ssl4gie_layernorm_bwd_workspace_bytes(rows, cols), dtype=torch.uint8,
                     device=x.device)
    _lib.check(L.ssl4gie_layernorm_bwd(ptr(dy), code(dy.dtype), ptr(x), ptr(gamma), ptr(mean),
                                       ptr(rstd), ptr(dres), ptr(dx), ptr(dx_lp), code(dy.dtype),
                                       ptr(dgamma), ptr(dbeta), int(accumulate), ptr(ws), rows,
                                       cols, stream()), "layernorm_bwd")
    return dx, dx_lp, dgamma, dbeta


def colsum(x2d, out=None, accumulate=False):
    _dev(x2d, out)
    rows, cols = x2d.shape
    L = _lib.load()
    if out is None:
        out = torch.empty(cols, dtype=torch.float32, device=x2d.device)
    assert out.numel() == cols and out.dtype == torch.float32
    ws = torch.empty(max(1, L.ssl4gie_colsum_workspace_bytes(rows, cols)), dtype=torch.uint8,
                     device=x2d.device)
    _lib.check(L.ssl4gie_colsum(ptr(x2d), code(x2d.dtype), ptr(out), int(accumulate), ptr(ws),
                                rows, cols, cols, stream()), "colsum")
    return out


# ------------------------------------------------------------------ GEMM
def gemm_raw(desc: GemmDesc, device):
    L = _lib.load()
    nbytes = L.ssl4gie_gemm_workspace_bytes(C.byref(desc))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=device) if nbytes else None
    _lib.check(L.ssl4gie_gemm(C.byref(desc), ptr(ws), nbytes, stream()), "gemm")


def _desc(M, N, K, dt_ab, dt_c):
    d = GemmDesc()
    d.M, d.N, d.K, d.batch1, d.batch2 = M, N, K, 1, 1
    d.dtype_ab, d.dtype_c, d.alpha, d.epilogue = dt_ab, dt_c, 1.0, _lib.EPI_NONE
    return d


def colstats_ok(T, n_out, k_in, dtype):
    """whether the producing GEMM can emit BatchNorm partial statistics (256x256 NT kernel rules)"""
    return dtype == torch.bfloat16 and k_in % 64 == 0 and n_out % 8 == 0 and T > 0


def linear_fwd(x2d, w, bias=None, out_dtype=None, epilogue=None, residual=None, colstats=False):
    """y[T, n_out] = x2d[T, k_in] @ w[n_out, k_in]^T (+bias | +bias+residual | gelu pair).
    colstats=True (bf16, no bias): also returns the per-128-row column sums / sums of squares of y
    ([ceil(T/128), 2, n_out] fp32) for the BatchNorm that follows."""
    _dev(x2d, w, bias, residual)
    T, k_in = x2d.shape
    n_out, k2 = w.shape
    assert k2 == k_in and x2d.dtype == w.dtype, (x2d.shape, w.shape, x2d.dtype, w.dtype)
    out_dtype = out_dtype or x2d.dtype
    d = _desc(T, n_out, k_in, code(x2d.dtype), code(out_dtype))
    d.A, d.sAm, d.sAk = ptr(x2d), k_in, 1
    d.B, d.sBk, d.sBn = ptr(w), 1, k_in
    y = torch.empty(T, n_out, dtype=out_dtype, device=x2d.device)
    d.C, d.ldc = ptr(y), n_out
    out2 = None
    if epilogue is None:
        epilogue = _lib.EPI_BIAS if bias is not None else _lib.EPI_NONE
    d.epilogue = epilogue
    if bias is not None:
        _f32(bias)
        assert bias.numel() == n_out
        d.bias = ptr(bias)
    if epilogue == _lib.EPI_BIAS_RESIDUAL:
        _f32(residual)
        assert residual.shape == (T, n_out)
        d.residual, d.ldr = ptr(residual), n_out
    if epilogue in (_lib.EPI_BIAS_GELU, _lib.EPI_BIAS_GELU_GRAD):
        out2 = torch.empty_like(y)  # (u, gelu(u)) resp. (gelu'(u), gelu(u))
        d.out2 = ptr(out2)
    if colstats:
        assert bias is None and epilogue == _lib.EPI_NONE and colstats_ok(T, n_out, k_in, x2d.dtype)
        stats = torch.empty((T + 127) // 128, 2, n_out, dtype=torch.float32, device=x2d.device)
        d.colstats = ptr(stats)
        gemm_raw(d, x2d.device)
        return y, stats
    gemm_raw(d, x2d.device)
    return (y, out2) if out2 is not None else y


def linear_colstats_only(x2d, w):
    """the BatchNorm partial statistics [ceil(T/128), 2, n_out] of y = x2d @ w^T (of its bf16-rounded values, as
    linear_fwd(colstats=True) returns them) WITHOUT writing y: the first half of the BatchNorm-fused 1x1
    convolution (linear_affine_fwd is the second)"""
    _dev(x2d, w)
    T, k_in = x2d.shape
    n_out = w.shape[0]
    assert w.shape[1] == k_in and x2d.dtype == w.dtype and colstats_ok(T, n_out, k_in, x2d.dtype)
    d = _desc(T, n_out, k_in, code(x2d.dtype), code(x2d.dtype))
    d.A, d.sAm, d.sAk = ptr(x2d), k_in, 1
    d.B, d.sBk, d.sBn = ptr(w), 1, k_in
    d.C, d.ldc = None, n_out
    d.epilogue = _lib.EPI_NONE
    stats = torch.empty((T + 127) // 128, 2, n_out, dtype=torch.float32, device=x2d.device)
    d.colstats = ptr(stats)
    gemm_raw(d, x2d.device)
    return stats


def linear_affine_fwd(x2d, w, scale, shift, aux=None, relu=False):
    """act((x2d @ w^T) * scale[n] + shift[n] (+ aux)), act = ReLU if `relu` (EPI_AFFINE_AUX_RELU): bf16 operands
    and output, the affine map applied to the fp32 accumulators"""
    _dev(x2d, w, scale, shift, aux)
    _f32(scale); _f32(shift)
    T, k_in = x2d.shape
    n_out = w.shape[0]
    assert w.shape[1] == k_in and x2d.dtype == w.dtype == torch.bfloat16
    assert scale.numel() == n_out and shift.numel() == n_out and scale.is_contiguous() and shift.is_contiguous()
    d = _desc(T, n_out, k_in, code(x2d.dtype), code(x2d.dtype))
    d.A, d.sAm, d.sAk = ptr(x2d), k_in, 1
    d.B, d.sBk, d.sBn = ptr(w), 1, k_in
    y = torch.empty(T, n_out, dtype=x2d.dtype, device=x2d.device)
    d.C, d.ldc = ptr(y), n_out
    d.epilogue, d.scale, d.bias, d.relu = _lib.EPI_AFFINE_AUX_RELU, ptr(scale), ptr(shift), int(bool(relu))
    if aux is not None:
        assert aux.shape == (T, n_out) and aux.dtype == x2d.dtype and aux.is_contiguous()
        d.aux = ptr(aux)
    gemm_raw(d, x2d.device)
    return y


def bn_coef_partials(partials, rows, gamma, beta, running_mean, running_var, momentum, eps):
    """training-mode BatchNorm statistics from GEMM-epilogue partials -> (coef [2, C]: y = x coef[0] + coef[1],
    mean, rstd); running statistics updated (ssl4gie_bn_coef_partials)"""
    _dev(partials, gamma, beta, running_mean, running_var)
    C = partials.shape[2]
    assert partials.dtype == torch.float32 and partials.shape[1] == 2 and partials.is_contiguous()
    L = _lib.load()
    dev = partials.device
    mean = torch.empty(C, dtype=torch.float32, device=dev)
    rstd = torch.empty(C, dtype=torch.float32, device=dev)
    coef = torch.empty(2, C, dtype=torch.float32, device=dev)
    ws = torch.empty(L.ssl4gie_bn_workspace_bytes(rows, C), dtype=torch.uint8, device=dev)
    _lib.check(L.ssl4gie_bn_coef_partials(ptr(partials), partials.shape[0], ptr(gamma), ptr(beta), ptr(mean),
                                          ptr(rstd), ptr(running_mean), ptr(running_var), float(momentum),
                                          float(eps), ptr(coef), ptr(ws), rows, C, stream()), "bn_coef_partials")
    return coef, mean, rstd


def add_aux_ok(T, k_in, n_out, dtype, has_wt):
    """whether linear_bwd_data can join a second gradient contribution in its epilogue"""
    return dtype == torch.bfloat16 and has_wt and n_out % 64 == 0 and k_in % 8 == 0 and T > 0


def linear_bwd_data(dy2d, w, w_t=None, out_dtype=None, dgelu_aux=None, mul_aux=None, add_aux=None):
    """dx[T, k_in] = dy[T, n_out] @ w[n_out, k_in]; uses w_t[k_in, n_out] (NT fast path) if given.
    `dgelu_aux` = u: dx *= gelu'(u);  `mul_aux` = m: dx *= m (m = gelu'(u) saved by the forward);
    `add_aux` = g: dx += g (another gradient contribution of the same tensor; see add_aux_ok)."""
    _dev(dy2d, w, w_t, dgelu_aux, mul_aux, add_aux)
    T, n_out = dy2d.shape
    k_in = w.shape[1]
    assert w.shape[0] == n_out
    out_dtype = out_dtype or dy2d.dtype
    d = _desc(T, k_in, n_out, code(dy2d.dtype), code(out_dtype))
    d.A, d.sAm, d.sAk = ptr(dy2d), n_out, 1
    if w_t is not None:
        assert w_t.shape == (k_in, n_out) and w_t.dtype == dy2d.dtype
        d.B, d.sBk, d.sBn = ptr(w_t), 1, n_out
    else:
        assert w.dtype == dy2d.dtype
        d.B, d.sBk, d.sBn = ptr(w), k_in, 1
    dx = torch.empty(T, k_in, dtype=out_dtype, device=dy2d.device)
    d.C, d.ldc = ptr(dx), k_in
    if dgelu_aux is not None:
        assert dgelu_aux.shape == dx.shape and dgelu_aux.dtype == out_dtype
        d.epilogue, d.aux = _lib.EPI_DGELU, ptr(dgelu_aux)
    if mul_aux is not None:
        assert dgelu_aux is None and mul_aux.shape == dx.shape and mul_aux.dtype == out_dtype
        d.epilogue, d.aux = _lib.EPI_MUL_AUX, ptr(mul_aux)
    if add_aux is not None:
        assert dgelu_aux is None and mul_aux is None and add_aux.dtype == out_dtype and \
            add_aux.numel() == dx.numel() and add_aux.is_contiguous() and \
            add_aux_ok(T, k_in, n_out, dy2d.dtype, w_t is not None)
        d.epilogue, d.aux = _lib.EPI_ADD_AUX, ptr(add_aux)
    gemm_raw(d, dy2d.device)
    return dx


def linear_bwd_weight(dy2d, x2d, out=None, accumulate=False, bias_out=None):
    """dW[n_out, k_in] = dy[T, n_out]^T @ x[T, k_in]  (fp32 output); with `bias_out` [n_out] the
    bias gradient (column sums of dy) is produced by the same call."""
    _dev(dy2d, x2d, out, bias_out)
    T, n_out = dy2d.shape
    T2, k_in = x2d.shape
    assert T == T2 and dy2d.dtype == x2d.dtype
    if out is None:
        out = torch.empty(n_out, k_in, dtype=torch.float32, device=x2d.device)
    assert out.dtype == torch.float32 and out.numel() == n_out * k_in
    d = _desc(n_out, k_in, T, code(dy2d.dtype), F32)
    d.A, d.sAm, d.sAk = ptr(dy2d), 1, n_out
    d.B, d.sBk, d.sBn = ptr(x2d), k_in, 1
    d.C, d.ldc = ptr(out), k_in
    d.accumulate = int(accumulate)
    if bias_out is not None:
        assert bias_out.dtype == torch.float32 and bias_out.numel() == n_out
        d.colsum_a = ptr(bias_out)
    gemm_raw(d, x2d.device)
    return out


def linear_bwd_weight_pair(dy_a, x_a, dy_b, x_b, bias_a=None, bias_b=None):
    """two weight-gradient products with the same token count in ONE launch (ssl4gie_gemm_tn_pair):
    (dW_a [n_a, k_a], dW_b [n_b, k_b]) = (dy_a^T x_a, dy_b^T x_b); optional fused bias gradients"""
    _dev(dy_a, x_a, dy_b, x_b, bias_a, bias_b)
    T = dy_a.shape[0]
    assert dy_b.shape[0] == T and x_a.shape[0] == T and x_b.shape[0] == T
    outs, descs = [], []
    for dy, x, b in ((dy_a, x_a, bias_a), (dy_b, x_b, bias_b)):
        n_out, k_in = dy.shape[1], x.shape[1]
        out = torch.empty(n_out, k_in, dtype=torch.float32, device=x.device)
        d = _desc(n_out, k_in, T, code(dy.dtype), F32)
        d.A, d.sAm, d.sAk = ptr(dy), 1, n_out
        d.B, d.sBk, d.sBn = ptr(x), k_in, 1
        d.C, d.ldc = ptr(out), k_in
        if b is not None:
            assert b.dtype == torch.float32 and b.numel() == n_out
            d.colsum_a = ptr(b)
        outs.append(out)
        descs.append(d)
    L = _lib.load()
    nbytes = L.ssl4gie_gemm_tn_pair_workspace_bytes(C.byref(descs[0]), C.byref(descs[1]))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dy_a.device) if nbytes else None
    _lib.check(L.ssl4gie_gemm_tn_pair(C.byref(descs[0]), C.byref(descs[1]), ptr(ws), nbytes, stream()),
               "gemm_tn_pair")
    return outs[0], outs[1]


def linear_bwd_weight_group(pairs, accumulate_into=None):
    """n weight-gradient products with the same token count in ONE launch (ssl4gie_gemm_tn_group).
    pairs: [(dy [T, n_i], x [T, k_i], bias_out or None), ...] -> [dW_i [n_i, k_i]].
    accumulate_into: optional list of existing fp32 dW tensors to accumulate into (C += dY^T X)."""
    T = pairs[0][0].shape[0]
    n = len(pairs)
    descs = (GemmDesc * n)()
    outs = []
    for i, (dy, x, b) in enumerate(pairs):
        _dev(dy, x, b)
        assert dy.shape[0] == T and x.shape[0] == T
        n_out, k_in = dy.shape[1], x.shape[1]
        if accumulate_into is not None:
            out = accumulate_into[i]
            assert out.shape == (n_out, k_in) and out.dtype == torch.float32
        else:
            out = torch.empty(n_out, k_in, dtype=torch.float32, device=x.device)
        d = descs[i]
        d.M, d.N, d.K, d.batch1, d.batch2 = n_out, k_in, T, 1, 1
        d.dtype_ab, d.dtype_c, d.alpha, d.epilogue = code(dy.dtype), F32, 1.0, _lib.EPI_NONE
        d.A, d.sAm, d.sAk = ptr(dy), 1, n_out
        d.B, d.sBk, d.sBn = ptr(x), k_in, 1
        d.C, d.ldc = ptr(out), k_in
        d.accumulate = 1 if accumulate_into is not None else 0
        if b is not None:
            assert b.dtype == torch.float32 and b.numel() == n_out
            d.colsum_a = ptr(b)
        outs.append(out)
    L = _lib.load()
    nbytes = L.ssl4gie_gemm_tn_group_workspace_bytes(descs, n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=pairs[0][0].device) if nbytes else None
    _lib.check(L.ssl4gie_gemm_tn_group(descs, n, ptr(ws), nbytes, stream()), "gemm_tn_group")
    return outs


# ------------------------------------------------------------------ attention
def attn_fwd(qkv, B, N, H, hd):
    _dev(qkv)
    D = H * hd
    assert qkv.numel() == B * N * 3 * D
    L = _lib.load()
    out = torch.empty(B, N, D, dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty(B, H, N, dtype=torch.float32, device=qkv.device)
    nb = L.ssl4gie_attn_workspace_bytes(code(qkv.dtype), B, N, H, hd)
    ws = torch.empty(nb, dtype=torch.uint8, device=qkv.device) if nb else None
    _lib.check(L.ssl4gie_attn_fwd(ptr(qkv), ptr(out), ptr(lse), code(qkv.dtype), B, N, H, hd,
                                  ptr(ws), stream()), "attn_fwd")
    return out, lse


def attn_bwd(qkv, out, dout, lse, B, N, H, hd):
    _dev(qkv, out, dout, lse)
    assert out.dtype == qkv.dtype and dout.dtype == qkv.dtype and lse.dtype == torch.float32
    assert out.numel() == B * N * H * hd and dout.numel() == out.numel()
    L = _lib.load()
    dqkv = torch.empty_like(qkv)
    nb = L.ssl4gie_attn_workspace_bytes(code(qkv.dtype), B, N, H, hd)
    ws = torch.empty(nb, dtype=torch.uint8, device=qkv.device) if nb else None
    _lib.check(L.ssl4gie_attn_bwd(ptr(qkv), ptr(out), ptr(dout), ptr(lse), ptr(dqkv),
                                  code(qkv.dtype), B, N, H, hd, ptr(ws), stream()), "attn_bwd")
    return dqkv


# ------------------------------------------------------------------ casts
def cast(src, dtype, out=None):
    _dev(src, out)
    _f32(src)
    if out is None:
        out = torch.empty(src.shape, dtype=dtype, device=src.device)
    assert out.numel() == src.numel() and out.dtype == dtype
    _lib.check(_lib.load().ssl4gie_cast(ptr(src), ptr(out), code(dtype), src.numel(), stream()),
               "cast")
    return out


def cast_transpose(src2d, dtype, out=None):
    _dev(src2d, out)
    _f32(src2d)
    r, c = src2d.shape
    if out is None:
        out = torch.empty(c, r, dtype=dtype, device=src2d.device)
    assert out.numel() == r * c and out.dtype == dtype
    _lib.check(_lib.load().ssl4gie_cast_transpose(ptr(src2d), ptr(out), code(dtype), r, c,
                                                  stream()), "cast_transpose")
    return out


def add_cast(a, b=None, want_f32=True, lp_dtype=None):
    """out = a + b on the fp32 gradient stream; optionally also its operand-type copy."""
    _dev(a, b)
    _f32(a, b)
    assert b is None or b.shape == a.shape
    out = torch.empty_like(a) if want_f32 else None
    out_lp = torch.empty(a.shape, dtype=lp_dtype, device=a.device) if lp_dtype is not None else None
    _lib.check(_lib.load().ssl4gie_add_cast(ptr(a), ptr(b), ptr(out), ptr(out_lp),
                                            code(lp_dtype) if lp_dtype is not None else 0,
                                            a.numel(), stream()), "add_cast")
    return out, out_lp


# ------------------------------------------------------------------ MAE glue
def mask_argsort(noise, len_keep):
    _dev(noise)
    _f32(noise)
    B, Lp = noise.shape
    ids_shuffle = torch.empty(B, Lp, dtype=torch.int64, device=noise.device)
    ids_restore = torch.empty_like(ids_shuffle)
    mask = torch.empty(B, Lp, dtype=torch.float32, device=noise.device)
    _lib.check(_lib.load().ssl4gie_mask_argsort(ptr(noise), ptr(ids_shuffle), ptr(ids_restore),
                                                ptr(mask), B, Lp, int(len_keep), stream()),
               "mask_argsort")
    return ids_shuffle, ids_restore, mask


def patch_gather(img, p, ids=None, nsel=None, out_dtype=torch.float32, order=0):
    _dev(img, ids)
    _f32(img)
    B, Cc, H, W = img.shape
    npatch = (H // p) * (W // p)
    if ids is not None:
        assert ids.dtype == torch.int64 and ids.shape[0] == B and ids.dim() == 2
        ids_stride = ids.shape[1]
        nsel = nsel if nsel is not None else ids.shape[1]
        assert nsel <= ids.shape[1]
    else:
        ids_stride, nsel = 0, npatch
    out = torch.empty(B * nsel, Cc * p * p, dtype=out_dtype, device=img.device)
    _lib.check(_lib.load().ssl4gie_patch_gather(ptr(img), ptr(ids), ptr(out), code(out_dtype), B,
                                                Cc, H, W, p, nsel, ids_stride, order, stream()),
               "patch_gather")
    return out


def tokens_assemble(y2d, cls, pos, B, nsel, ids=None):
    _dev(y2d, cls, pos, ids)
    _f32(cls, pos)
    D = y2d.shape[1]
    assert y2d.shape[0] == B * nsel and cls.numel() == D and pos.shape[-1] == D
    ids_stride = 0
    if ids is not None:
        assert ids.dtype == torch.int64 and ids.shape[0] == B and ids.shape[1] >= nsel
        ids_stride = ids.shape[1]
        assert pos.numel() // D >= 1 + ids.shape[1]
    else:
        assert pos.numel() // D >= 1 + nsel
    x = torch.empty(B, nsel + 1, D, dtype=torch.float32, device=y2d.device)
    _lib.check(_lib.load().ssl4gie_tokens_assemble(ptr(y2d), code(y2d.dtype), ptr(cls), ptr(pos),
                                                   ptr(ids), ids_stride, ptr(x), B, nsel, D,
                                                   stream()), "tokens_assemble")
    return x


def tokens_assemble_bwd(dx, lp_dtype, dcls_out=None, accumulate=False):
    """dy[b*nsel+j] = dx[b,1+j] (operand type); dcls_out (+)= sum_b dx[b,0] if given."""
    _dev(dx, dcls_out)
    _f32(dx, dcls_out)
    B, n1, D = dx.shape
    assert dcls_out is None or dcls_out.numel() == D
    dy = torch.empty(B * (n1 - 1), D, dtype=lp_dtype, device=dx.device)
    _lib.check(_lib.load().ssl4gie_tokens_assemble_bwd(ptr(dx), ptr(dy), code(lp_dtype),
                                                       ptr(dcls_out), int(accumulate), B, n1 - 1,
                                                       D, stream()), "tokens_assemble_bwd")
    return dy


def decoder_assemble(y, mask_token, dpos, ids_restore, nkeep):
    _dev(y, mask_token, dpos, ids_restore)
    _f32(mask_token, dpos)
    B, Lp = ids_restore.shape
    D = y.shape[-1]
    assert y.numel() == B * (nkeep + 1) * D and mask_token.numel() == D
    assert dpos.numel() == (Lp + 1) * D and ids_restore.dtype == torch.int64
    xd = torch.empty(B, Lp + 1, D, dtype=torch.float32, device=y.device)
    _lib.check(_lib.load().ssl4gie_decoder_assemble(ptr(y), code(y.dtype), ptr(mask_token),
                                                    ptr(dpos), ptr(ids_restore), ptr(xd), B, Lp,
                                                    nkeep, D, stream()), "decoder_assemble")
    return xd


def decoder_assemble_bwd(dxd, ids_shuffle, nkeep, lp_dtype, dmask_out, accumulate=False):
    _dev(dxd, ids_shuffle, dmask_out)
    _f32(dxd, dmask_out)
    B, L1, D = dxd.shape
    Lp = L1 - 1
    assert ids_shuffle.shape == (B, Lp) and ids_shuffle.dtype == torch.int64
    assert dmask_out.numel() == D
    L = _lib.load()
    dy = torch.empty(B, nkeep + 1, D, dtype=lp_dtype, device=dxd.device)
    ws = torch.empty(L.ssl4gie_decoder_assemble_bwd_workspace_bytes(B, Lp, D), dtype=torch.uint8,
                     device=dxd.device)
    _lib.check(L.ssl4gie_decoder_assemble_bwd(ptr(dxd), ptr(ids_shuffle), ptr(dy), code(lp_dtype),
                                              ptr(dmask_out), int(accumulate), ptr(ws), B, Lp,
                                              nkeep, D, stream()), "decoder_assemble_bwd")
    return dy


def _loss_geom(pred, img, mask, p):
    B, Cc, H, W = img.shape
    Lp = (H // p) * (W // p)
    assert mask.shape == (B, Lp) and pred.dim() == 3 and pred.shape[0] == B
    assert pred.shape[2] == Cc * p * p and pred.shape[1] in (Lp, Lp + 1)
    return B, Cc, H, W, int(pred.shape[1] == Lp + 1)


def mae_loss_fwd(pred, img, mask, p, norm_pix):
    """per_patch[B, L] = mask * mean((pred - target)^2); pred is [B, L, P] or [B, 1+L, P]."""
    _dev(pred, img, mask)
    _f32(pred, img, mask)
    B, Cc, H, W, has_cls = _loss_geom(pred, img, mask, p)
    per_patch = torch.empty_like(mask)
    _lib.check(_lib.load().ssl4gie_mae_loss(ptr(pred), ptr(img), ptr(mask), ptr(per_patch), 0, 0,
                                            1.0, int(norm_pix), has_cls, B, Cc, H, W, p, stream()),
               "mae_loss")
    return per_patch


def mae_loss_bwd(pred, img, mask, p, norm_pix, gpp, gscale_host=1.0):
    _dev(pred, img, mask, gpp)
    _f32(pred, img, mask, gpp)
    assert gpp is None or gpp.shape == mask.shape
    B, Cc, H, W, has_cls = _loss_geom(pred, img, mask, p)
    dpred = torch.empty_like(pred)
    _lib.check(_lib.load().ssl4gie_mae_loss(ptr(pred), ptr(img), ptr(mask), 0, ptr(dpred),
                                            ptr(gpp), float(gscale_host), int(norm_pix), has_cls,
                                            B, Cc, H, W, p, stream()), "mae_loss(bwd)")
    return dpred


# ------------------------------------------------------------------ DPT decoder glue (channels-last)
def _nhwc(x):
    _dev(x)
    assert x.dim() == 4, "expected [B, H, W, C]"
    return x.shape


def conv_out_hw(H, W, stride):
    return (H - 1) // stride + 1, (W - 1) // stride + 1


def k_pad(k, dtype):
    """row stride of a patch matrix: the bf16 MFMA GEMM wants whole 64-deep K-tiles"""
    return (k + 63) // 64 * 64 if dtype == torch.bfloat16 else k


def im2col3x3(x, stride=1, relu=False, ld=None):
    B, H, W, C = _nhwc(x)
    Ho, Wo = conv_out_hw(H, W, stride)
    ld = ld or k_pad(9 * C, x.dtype)
    cols = torch.empty(B * Ho * Wo, ld, dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().ssl4gie_im2col3x3(ptr(x), ptr(cols), code(x.dtype), B, H, W, C, stride,
                                             int(relu), ld, stream()), "im2col3x3")
    return cols


def conv3x3_implicit_ok(x, stride, n_out=None, wgrad=False):
    """whether the gathered 256x256 GEMM kernels take this map (ssl4gie_conv3x3_geom limits)"""
    if x.dtype != torch.bfloat16 or not x.is_contiguous():
        return False
    B, H, W, C = x.shape
    Ho, Wo = conv_out_hw(H, W, stride)
    if stride not in (1, 2) or Wo < 2 or (B * H + 1) * W * C * 2 >= 2 ** 31:
        return False
    if wgrad:  # contraction over pixels in whole K-tiles; dy rows addressed with 32-bit offsets
        return C % 8 == 0 and (B * Ho * Wo) % 64 == 0 and n_out % 8 == 0 and \
            B * Ho * Wo * n_out * 2 < 2 ** 32
    return C % 64 == 0 and n_out % 8 == 0


def _geom(x, stride, relu):
    B, H, W, C_ = x.shape
    g = _lib.Conv3x3Geom()
    g.B, g.H, g.W, g.C, g.stride, g.relu = B, H, W, C_, stride, int(relu)
    return g


def conv3x3_fwd(x, w2, bias=None, stride=1, relu=False, relu_mask=None, colstats=False):
    """y[(b,oy,ox), co] = sum_k P[(b,oy,ox), k] w2[co, k] (+ bias), P = implicit patch matrix of the
    bf16 map x [B,H,W,C] (never materialised; ssl4gie_gemm_desc.conv), w2 [Cout, 9C].
    `relu_mask` [B,Ho,Wo,Cout]: y = relu_mask > 0 ? y : 0 in the epilogue (the data gradient of a
    convolution whose input went through a ReLU; no bias, no relu)."""
    _dev(x, w2, bias, relu_mask)
    B, H, W, Cin = _nhwc(x)
    Cout, K = w2.shape
    assert K == 9 * Cin and w2.dtype == x.dtype and w2.is_contiguous()
    assert conv3x3_implicit_ok(x, stride, Cout)
    Ho, Wo = conv_out_hw(H, W, stride)
    M = B * Ho * Wo
    d = _desc(M, Cout, K, code(x.dtype), code(x.dtype))
    g = _geom(x, stride, relu)
    d.conv = C.pointer(g)
    d.A, d.sAm, d.sAk = ptr(x), K, 1
    d.B, d.sBk, d.sBn = ptr(w2), 1, K
    y = torch.empty(M, Cout, dtype=x.dtype, device=x.device)
    d.C, d.ldc = ptr(y), Cout
    if bias is not None:
        _f32(bias)
        assert bias.numel() == Cout
        d.epilogue, d.bias = _lib.EPI_BIAS, ptr(bias)
    if relu_mask is not None:
        assert bias is None and not relu and relu_mask.dtype == x.dtype and relu_mask.is_contiguous() \
            and relu_mask.numel() == M * Cout
        d.epilogue, d.aux = _lib.EPI_RELU_MASK_AUX, ptr(relu_mask)
    if colstats:  # BatchNorm partial statistics of y (see linear_fwd)
        assert bias is None and relu_mask is None
        stats = torch.empty((M + 127) // 128, 2, Cout, dtype=torch.float32, device=x.device)
        d.colstats = ptr(stats)
        gemm_raw(d, x.device)
        return y.view(B, Ho, Wo, Cout), stats
    gemm_raw(d, x.device)
    return y.view(B, Ho, Wo, Cout)


def conv3x3_bwd_weight(dy2d, x, stride=1, relu=False, bias_out=None):
    """dW2[co, k] = sum_pixels dy[(b,oy,ox), co] P[(b,oy,ox), k] (fp32 [Cout, 9C]); with `bias_out`
    the bias gradient rides on the same product."""
    _dev(dy2d, x, bias_out)
    B, H, W, Cin = _nhwc(x)
    T, Cout = dy2d.shape
    Ho, Wo = conv_out_hw(H, W, stride)
    assert T == B * Ho * Wo and dy2d.dtype == x.dtype and dy2d.is_contiguous()
    assert conv3x3_implicit_ok(x, stride, Cout, wgrad=True)
    out = torch.empty(Cout, 9 * Cin, dtype=torch.float32, device=x.device)
    d = _desc(Cout, 9 * Cin, T, code(x.dtype), F32)
    g = _geom(x, stride, relu)
    d.conv = C.pointer(g)
    d.A, d.sAm, d.sAk = ptr(dy2d), 1, Cout
    d.B, d.sBk, d.sBn = ptr(x), 9 * Cin, 1
    d.C, d.ldc = ptr(out), 9 * Cin
    if bias_out is not None:
        assert bias_out.dtype == torch.float32 and bias_out.numel() == Cout
        d.colsum_a = ptr(bias_out)
    gemm_raw(d, x.device)
    return out


_DIRECT_SMALL = os.environ.get("SSL4GIE_DIRECT_SMALL", "1") != "0"


def conv3x3_direct_supported(x, n_out):
    """whether the direct kernel CAN take this stride-1 map (bf16, Cin % 32 == 0, Cout % 8 == 0)"""
    if x.dtype != torch.bfloat16 or not x.is_contiguous():
        return False
    B, H, W, Cin = x.shape
    return bool(_lib.load().ssl4gie_conv3x3_direct_ok(B, H, W, Cin, n_out))


def conv3x3_direct_ok(x, n_out):
    """whether the direct (halo-in-LDS) kernel takes this map: narrow layers the 256-wide GEMM tiles
    would mostly pad (ssl4gie_conv3x3_direct_fwd)"""
    if x.dtype != torch.bfloat16 or not x.is_contiguous():
        return False
    B, H, W, Cin = x.shape
    # measured (tools/conv_bench.py): ahead of the gathered GEMM wherever its 256-wide tiles are
    # mostly padding — narrow layers — and on small maps (<= 16 wide: 256 -> 256 @14, 512 -> 512 @7),
    # where the GEMM has too few tiles to fill the chip
    return (n_out <= 128 or Cin == 32 or (W <= 16 and _DIRECT_SMALL)) and \
        bool(_lib.load().ssl4gie_conv3x3_direct_ok(B, H, W, Cin, n_out))


def conv3x3_direct_fwd(x, w2, bias=None, relu=False, relu_mask=None, colstats=False, in_coef=None):
    """stride-1 3x3 convolution of the bf16 map x [B,H,W,Cin] with w2 [Cout, 9 Cin] on the direct
    kernel; semantics of conv3x3_fwd (bias and relu_mask may be combined with relu here).
    colstats: also return the BatchNorm partial statistics [tiles, 2, Cout] of y.
    in_coef [2, Cin] (ops.bn_coef_partials): the operand is act(x in_coef[0] + in_coef[1]) — a BatchNorm (+ ReLU)
    applied on the way in, zero padding after it (ssl4gie_conv3x3_direct_fwd_affine)."""
    _dev(x, w2, bias, relu_mask, in_coef)
    B, H, W, Cin = _nhwc(x)
    Cout, K = w2.shape
    assert K == 9 * Cin and w2.dtype == x.dtype == torch.bfloat16 and w2.is_contiguous() and x.is_contiguous()
    if bias is not None:
        _f32(bias)
        assert bias.numel() == Cout
    if relu_mask is not None:
        assert relu_mask.dtype == x.dtype and relu_mask.is_contiguous() and relu_mask.numel() == B * H * W * Cout
    y = torch.empty(B, H, W, Cout, dtype=x.dtype, device=x.device)
    L = _lib.load()
    stats = None
    if colstats:
        assert relu_mask is None
        stats = torch.empty(L.ssl4gie_conv3x3_direct_tiles(B, H, W), 2, Cout, dtype=torch.float32, device=x.device)
    if in_coef is not None:
        _f32(in_coef)
        assert relu_mask is None and in_coef.shape == (2, Cin) and in_coef.is_contiguous()
        _lib.check(L.ssl4gie_conv3x3_direct_fwd_affine(ptr(x), ptr(in_coef), ptr(w2), ptr(bias), ptr(y), ptr(stats),
                                                       B, H, W, Cin, Cout, int(relu), stream()),
                   "conv3x3_direct_fwd_affine")
        return (y, stats) if colstats else y
    _lib.check(L.ssl4gie_conv3x3_direct_fwd(ptr(x), ptr(w2), ptr(bias), ptr(relu_mask), ptr(y), ptr(stats),
                                            B, H, W, Cin, Cout, int(relu), stream()), "conv3x3_direct_fwd")
    return (y, stats) if colstats else y


def conv3x3_direct_wgrad_ok(x, n_out):
    """narrow layers only (n_out <= 128): from 256 couts on the gathered TN GEMM has full tiles"""
    if x.dtype != torch.bfloat16 or not x.is_contiguous():
        return False
    B, H, W, Cin = x.shape
    return n_out <= 128 and bool(_lib.load().ssl4gie_conv3x3_direct_wgrad_ok(B, H, W, Cin, n_out))


def conv3x3_direct_wgrad(dy, x, relu=False, bias_out=None, in_coef=None):
    """dW2 [Cout, 9 Cin] fp32 of the direct convolution: dy [B,H,W,Cout] (or [B*H*W, Cout], Cout % 32 == 0),
    x [B,H,W,Cin]; with `bias_out` [Cout] fp32 the bias gradient is produced by the same kernel;
    in_coef: as conv3x3_direct_fwd (the operand of the forward is rebuilt on the way in)"""
    _dev(dy, x, bias_out, in_coef)
    if bias_out is not None:
        assert bias_out.dtype == torch.float32 and bias_out.numel() == dy.shape[-1]
    B, H, W, Cin = _nhwc(x)
    Cout = dy.shape[-1]
    assert dy.numel() == B * H * W * Cout and dy.dtype == x.dtype and dy.is_contiguous()
    lib = _lib.load()
    nbytes = lib.ssl4gie_conv3x3_direct_wgrad_workspace_bytes(B, H, W, Cin, Cout)
    assert nbytes > 0, "conv3x3_direct_wgrad: unsupported geometry"
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    out = torch.empty(Cout, 9 * Cin, dtype=torch.float32, device=x.device)
    if in_coef is not None:
        _f32(in_coef)
        assert in_coef.shape == (2, Cin) and in_coef.is_contiguous()
        _lib.check(lib.ssl4gie_conv3x3_direct_wgrad_affine(ptr(dy), ptr(x), ptr(in_coef), ptr(out), ptr(bias_out), ptr(ws),
                                                           nbytes, B, H, W, Cin, Cout, int(relu), 0, stream()),
                   "conv3x3_direct_wgrad_affine")
        return out
    _lib.check(lib.ssl4gie_conv3x3_direct_wgrad(ptr(dy), ptr(x), ptr(out), ptr(bias_out), ptr(ws), nbytes, B, H, W, Cin,
                                                Cout, int(relu), 0, stream()), "conv3x3_direct_wgrad")
    return out


def stem7x7_pack(imgs):
    """fp32 NCHW [B,3,H,W] -> the padded 4-channel bf16 image the direct stem kernels read"""
    _dev(imgs)
    _f32(imgs)
    B, Cc, H, W = imgs.shape
    assert Cc == 3 and imgs.is_contiguous()
    L = _lib.load()
    nb = L.ssl4gie_stem7x7_packed_bytes(B, H, W)
    assert nb > 0, "stem7x7: unsupported image size"
    out = torch.empty(nb // 2, dtype=torch.bfloat16, device=imgs.device)
    _lib.check(L.ssl4gie_stem7x7_pack(ptr(imgs), ptr(out), B, H, W, stream()), "stem7x7_pack")
    return out


def conv3x3_weight_pack(weight, dtype, mode, ld=None):
    """operand image of a Conv2d(k=3) weight [Cout, Cin, 3, 3] fp32 in one launch (cast included):
    mode 0 -> [Cout, ld] rows (tap, ci); mode 1 -> [Cin, ld] rows (flipped tap, co); mode 2 -> [ld, Cout]"""
    _dev(weight)
    assert weight.dtype == torch.float32 and weight.dim() == 4 and weight.shape[2:] == (3, 3)
    w = weight.contiguous()
    Cout, Cin = w.shape[:2]
    need = 9 * (Cout if mode == 1 else Cin)
    ld = need if ld is None else ld
    shape = (Cout, ld) if mode == 0 else ((Cin, ld) if mode == 1 else (ld, Cout))
    out = torch.empty(shape, dtype=dtype, device=w.device)
    _lib.check(_lib.load().ssl4gie_conv3x3_weight_pack(ptr(w), ptr(out), code(dtype), Cout, Cin, mode, ld, stream()),
               "conv3x3_weight_pack")
    return out


def conv3x3_weight_pack_batch(weights, dtype, modes, lds):
    """conv3x3_weight_pack for a list of weights in one launch (ssl4gie_conv3x3_weight_pack_batch); lds[i] None =
    the unpadded row length"""
    import ctypes as C
    n = len(weights)
    if n == 0:
        return []
    outs, geo = [], []
    for w, mode, ld in zip(weights, modes, lds):
        _dev(w)
        assert w.dtype == torch.float32 and w.dim() == 4 and w.shape[2:] == (3, 3) and w.is_contiguous()
        Cout, Cin = w.shape[:2]
        need = 9 * (Cout if mode == 1 else Cin)
        ld = need if ld is None else ld
        shape = (Cout, ld) if mode == 0 else ((Cin, ld) if mode == 1 else (ld, Cout))
        outs.append(torch.empty(shape, dtype=dtype, device=w.device))
        geo.append((Cout, Cin, mode, ld))
    vp_arr = C.c_void_p * n
    i_arr = C.c_int * n
    _lib.check(_lib.load().ssl4gie_conv3x3_weight_pack_batch(
        vp_arr(*[ptr(w) for w in weights]), vp_arr(*[ptr(o) for o in outs]), i_arr(*[g[0] for g in geo]),
        i_arr(*[g[1] for g in geo]), i_arr(*[g[2] for g in geo]), i_arr(*[g[3] for g in geo]), n, code(dtype),
        stream()), "conv3x3_weight_pack_batch")
    return outs


def conv3x3_wgrad_unpack(dw2, target, accumulate=False):
    """dw2 [Cout, ld >= 9 Cin] fp32, columns (tap, ci) -> (+)= target [Cout, Cin, 3, 3] (contiguous) in one launch"""
    _dev(dw2, target)
    Cout, Cin = target.shape[:2]
    assert dw2.dtype == torch.float32 and target.dtype == torch.float32 and target.is_contiguous()
    assert dw2.stride(1) == 1 and dw2.shape[0] == Cout and dw2.shape[1] >= 9 * Cin
    _lib.check(_lib.load().ssl4gie_conv3x3_wgrad_unpack(ptr(dw2), ptr(target), Cout, Cin, dw2.stride(0),
                                                         int(accumulate), stream()), "conv3x3_wgrad_unpack")
    return target


def stem7x7_weight(weight):
    """[64, 3, 7, 7] fp32 -> [64, 256] in the kernels' layout [co][ky (8)][kx (8)][c (4)], zeros in the padding"""
    w = torch.zeros(weight.shape[0], 8, 8, 4, dtype=weight.dtype, device=weight.device)
    w[:, :7, :7, :3] = weight.permute(0, 2, 3, 1)
    return w.reshape(weight.shape[0], 256)


def stem7x7_fwd(packed, w2s, B, H, W, colstats=False):
    """y [B,Ho,Wo,64] bf16 (+ BatchNorm partial statistics [tiles, 2, 64])"""
    _dev(packed, w2s)
    assert w2s.shape == (64, 256) and w2s.dtype == torch.bfloat16 and w2s.is_contiguous()
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    L = _lib.load()
    y = torch.empty(B, Ho, Wo, 64, dtype=torch.bfloat16, device=packed.device)
    stats = torch.empty(L.ssl4gie_stem7x7_tiles(B, H, W), 2, 64, dtype=torch.float32, device=packed.device) \
        if colstats else None
    _lib.check(L.ssl4gie_stem7x7_fwd(ptr(packed), ptr(w2s), ptr(y), ptr(stats), B, H, W, stream()), "stem7x7_fwd")
    return (y, stats) if colstats else y


def stem7x7_wgrad(dy, packed, B, H, W):
    """weight gradient [64, 3, 7, 7] fp32 of the stem from dy [B,Ho,Wo,64] bf16 and the packed image"""
    _dev(dy, packed)
    assert dy.dtype == torch.bfloat16 and dy.is_contiguous() and dy.shape[-1] == 64
    L = _lib.load()
    nb = L.ssl4gie_stem7x7_wgrad_workspace_bytes(B, H, W)
    ws = torch.empty(nb, dtype=torch.uint8, device=dy.device)
    out = torch.empty(64, 7, 8, 4, dtype=torch.float32, device=dy.device)
    _lib.check(L.ssl4gie_stem7x7_wgrad(ptr(dy), ptr(packed), ptr(out), ptr(ws), nb, B, H, W, 0, stream()),
               "stem7x7_wgrad")
    return out[:, :, :7, :3].permute(0, 3, 1, 2)  # [co, c, ky, kx]


def col2im3x3(dcols, B, H, W, C, stride):
    _dev(dcols)
    dx = torch.empty(B, H, W, C, dtype=dcols.dtype, device=dcols.device)
    _lib.check(_lib.load().ssl4gie_col2im3x3(ptr(dcols), ptr(dx), code(dcols.dtype), B, H, W, C,
                                             stride, dcols.shape[1], stream()), "col2im3x3")
    return dx


def bilinear2x_fwd(x):
    B, H, W, C = _nhwc(x)
    y = torch.empty(B, 2 * H, 2 * W, C, dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().ssl4gie_bilinear2x_fwd(ptr(x), ptr(y), code(x.dtype), B, H, W, C,
                                                  stream()), "bilinear2x_fwd")
    return y


def bilinear2x_bwd(dy):
    B, Ho, Wo, C = _nhwc(dy)
    dx = torch.empty(B, Ho // 2, Wo // 2, C, dtype=dy.dtype, device=dy.device)
    _lib.check(_lib.load().ssl4gie_bilinear2x_bwd(ptr(dy), ptr(dx), code(dy.dtype), B, Ho // 2,
                                                  Wo // 2, C, stream()), "bilinear2x_bwd")
    return dx


def pixel_shuffle(g, bias, B, H, W, k, C):
    _dev(g, bias)
    assert g.shape == (B * H * W, k * k * C)
    y = torch.empty(B, k * H, k * W, C, dtype=g.dtype, device=g.device)
    _lib.check(_lib.load().ssl4gie_pixel_shuffle(ptr(g), ptr(bias), ptr(y), code(g.dtype), B, H, W,
                                                 k, C, stream()), "pixel_shuffle")
    return y


def pixel_unshuffle(dy, k):
    B, Ho, Wo, C = _nhwc(dy)
    H, W = Ho // k, Wo // k
    dg = torch.empty(B * H * W, k * k * C, dtype=dy.dtype, device=dy.device)
    _lib.check(_lib.load().ssl4gie_pixel_unshuffle(ptr(dy), ptr(dg), code(dy.dtype), B, H, W, k, C,
                                                   stream()), "pixel_unshuffle")
    return dg


def tokens_to_map(z, dtype):
    _dev(z)
    _f32(z)
    B, L1, D = z.shape
    x = torch.empty(B * (L1 - 1), D, dtype=dtype, device=z.device)
    _lib.check(_lib.load().ssl4gie_tokens_to_map(ptr(z), ptr(x), code(dtype), B, L1 - 1, D, stream()),
               "tokens_to_map")
    return x


def map_to_tokens(dx, B, L, D):
    _dev(dx)
    dz = torch.empty(B, L + 1, D, dtype=torch.float32, device=dx.device)
    _lib.check(_lib.load().ssl4gie_map_to_tokens(ptr(dx), ptr(dz), code(dx.dtype), B, L, D, stream()),
               "map_to_tokens")
    return dz


def eltwise_add(a, b):
    _dev(a, b)
    assert a.shape == b.shape and a.dtype == b.dtype
    out = torch.empty_like(a)
    _lib.check(_lib.load().ssl4gie_eltwise(0, ptr(a), ptr(b), None, ptr(out), code(a.dtype), a.numel(),
                                           stream()), "eltwise add")
    return out


def relu_bwd(x, g, skip=None):
    """(x > 0 ? g : 0) + skip"""
    _dev(x, g, skip)
    assert x.shape == g.shape and x.dtype == g.dtype
    out = torch.empty_like(g)
    _lib.check(_lib.load().ssl4gie_eltwise(1, ptr(x), ptr(g), ptr(skip), ptr(out), code(g.dtype),
                                           g.numel(), stream()), "relu_bwd")
    return out


def depth_head_fwd(x2d, w, bias):
    _dev(x2d, w, bias)
    _f32(w, bias)
    M, C = x2d.shape
    y = torch.empty(M, dtype=torch.float32, device=x2d.device)
    _lib.check(_lib.load().ssl4gie_depth_head_fwd(ptr(x2d), ptr(w), ptr(bias), ptr(y), code(x2d.dtype),
                                                  M, C, stream()), "depth_head_fwd")
    return y


def depth_head_bwd(x2d, w, y, dy, dw, db, accumulate):
    _dev(x2d, w, y, dy, dw, db)
    _f32(w, y, dy, dw, db)
    M, C = x2d.shape
    L = _lib.load()
    ws = torch.empty(L.ssl4gie_depth_head_bwd_workspace_bytes(M, C), dtype=torch.uint8,
                     device=x2d.device)
    dx = torch.empty_like(x2d)
    _lib.check(L.ssl4gie_depth_head_bwd(ptr(x2d), ptr(w), ptr(y), ptr(dy), ptr(dx), ptr(dw), ptr(db),
                                        int(accumulate), ptr(ws), code(x2d.dtype), M, C, stream()),
               "depth_head_bwd")
    return dx


# ------------------------------------------------------------------ ResNet glue (channels-last)
def stem_im2col7x7(imgs, dtype):
    _dev(imgs)
    _f32(imgs)
    B, C, H, W = imgs.shape
    assert C == 3
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    ld = k_pad(147, dtype) if dtype == torch.bfloat16 else 152  # multiple of 8 columns
    cols = torch.empty(B * Ho * Wo, ld, dtype=dtype, device=imgs.device)
    _lib.check(_lib.load().ssl4gie_stem_im2col7x7(ptr(imgs), ptr(cols), code(dtype), B, H, W, ld,
                                                  stream()), "stem_im2col7x7")
    return cols, Ho, Wo


def subsample2(x):
    B, H, W, C = _nhwc(x)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty(B, Ho, Wo, C, dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().ssl4gie_subsample2(ptr(x), ptr(y), code(x.dtype), B, H, W, C, 0, stream()),
               "subsample2")
    return y


def subsample2_bwd(dy, H, W):
    B, Ho, Wo, C = _nhwc(dy)
    dx = torch.empty(B, H, W, C, dtype=dy.dtype, device=dy.device)
    _lib.check(_lib.load().ssl4gie_subsample2(ptr(dy), ptr(dx), code(dy.dtype), B, H, W, C, 1, stream()),
               "subsample2_bwd")
    return dx


def bn_fwd(x2d, gamma, beta, res, running_mean, running_var, momentum, eps, relu, training,
           mean=None, rstd=None, partials=None):
    """`partials` [parts, 2, C]: training-mode statistics from the producing GEMM's epilogue
    (linear_fwd / conv3x3_fwd with colstats=True) instead of a pass over x2d"""
    _dev(x2d, gamma, beta, res, running_mean, running_var, mean, rstd, partials)
    rows, C = x2d.shape
    L = _lib.load()
    y = torch.empty_like(x2d)
    if training:
        mean = torch.empty(C, dtype=torch.float32, device=x2d.device)
        rstd = torch.empty(C, dtype=torch.float32, device=x2d.device)
    ws = torch.empty(L.ssl4gie_bn_workspace_bytes(rows, C), dtype=torch.uint8, device=x2d.device)
    if training and partials is not None:
        assert partials.dtype == torch.float32 and partials.shape[1:] == (2, C) and partials.is_contiguous()
        _lib.check(L.ssl4gie_bn_fwd_partials(ptr(x2d), ptr(partials), partials.shape[0], ptr(gamma),
                                             ptr(beta), ptr(res), ptr(y), ptr(mean), ptr(rstd),
                                             ptr(running_mean), ptr(running_var), float(momentum),
                                             float(eps), int(relu), ptr(ws), code(x2d.dtype), rows, C,
                                             stream()), "bn_fwd_partials")
        return y, mean, rstd
    _lib.check(L.ssl4gie_bn_fwd(ptr(x2d), ptr(gamma), ptr(beta), ptr(res), ptr(y), ptr(mean), ptr(rstd),
                                ptr(running_mean), ptr(running_var), float(momentum), float(eps),
                                int(relu), int(training), ptr(ws), code(x2d.dtype), rows, C, stream()),
               "bn_fwd")
    return y, mean, rstd


def bn_bwd(dy2d, y2d, x2d, gamma, mean, rstd, relu, want_dres, dgamma, dbeta, accumulate):
    _dev(dy2d, y2d, x2d, gamma, mean, rstd, dgamma, dbeta)
    rows, C = x2d.shape
    L = _lib.load()
    dx = torch.empty_like(x2d)
    dres = torch.empty_like(x2d) if want_dres else None
    ws = torch.empty(L.ssl4gie_bn_workspace_bytes(rows, C), dtype=torch.uint8, device=x2d.device)
    _lib.check(L.ssl4gie_bn_bwd(ptr(dy2d), ptr(y2d), ptr(x2d), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx),
                                ptr(dres), ptr(dgamma), ptr(dbeta), int(accumulate), int(relu), ptr(ws),
                                code(x2d.dtype), rows, C, stream()), "bn_bwd")
    return dx, dres


def bn_fwd_bits(x2d, gamma, beta, res, running_mean, running_var, momentum, eps, partials):
    """training-mode BatchNorm (+ residual) + ReLU from GEMM-epilogue partials that also writes the ReLU mask as
    a bit map [rows * C / 8] (bf16): -> (y, bits, mean, rstd) (ssl4gie_bn_fwd_partials_bits)"""
    _dev(x2d, gamma, beta, res, running_mean, running_var, partials)
    rows, C = x2d.shape
    assert x2d.dtype == torch.bfloat16 and partials.dtype == torch.float32 and partials.shape[1:] == (2, C)
    L = _lib.load()
    y = torch.empty_like(x2d)
    bits = torch.empty(rows * C // 8, dtype=torch.uint8, device=x2d.device)
    mean = torch.empty(C, dtype=torch.float32, device=x2d.device)
    rstd = torch.empty(C, dtype=torch.float32, device=x2d.device)
    ws = torch.empty(L.ssl4gie_bn_workspace_bytes(rows, C), dtype=torch.uint8, device=x2d.device)
    _lib.check(L.ssl4gie_bn_fwd_partials_bits(ptr(x2d), ptr(partials), partials.shape[0], ptr(gamma), ptr(beta),
                                              ptr(res), ptr(y), ptr(bits), ptr(mean), ptr(rstd), ptr(running_mean),
                                              ptr(running_var), float(momentum), float(eps), ptr(ws),
                                              code(x2d.dtype), rows, C, stream()), "bn_fwd_partials_bits")
    return y, bits, mean, rstd


def bn_bwd_bits(dy2d, bits, x2d, gamma, mean, rstd, dgamma, dbeta, accumulate):
    """backward of bn_fwd_bits: -> (dx, dres = the masked gradient) (ssl4gie_bn_bwd_bits)"""
    _dev(dy2d, bits, x2d, gamma, mean, rstd, dgamma, dbeta)
    rows, C = x2d.shape
    assert bits.dtype == torch.uint8 and bits.numel() == rows * C // 8
    L = _lib.load()
    dx = torch.empty_like(x2d)
    dres = torch.empty_like(x2d)
    ws = torch.empty(L.ssl4gie_bn_workspace_bytes(rows, C), dtype=torch.uint8, device=x2d.device)
    _lib.check(L.ssl4gie_bn_bwd_bits(ptr(dy2d), ptr(bits), ptr(x2d), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx),
                                     ptr(dres), ptr(dgamma), ptr(dbeta), int(accumulate), ptr(ws), code(x2d.dtype),
                                     rows, C, stream()), "bn_bwd_bits")
    return dx, dres


def bn_bwd_xmask(dy2d, x2d, gamma, beta, mean, rstd, dgamma, dbeta, accumulate):
    """BatchNorm + ReLU without a residual input: the mask is rebuilt from x and the forward's coefficients, the
    ReLU output is not read (ssl4gie_bn_bwd_xmask)"""
    _dev(dy2d, x2d, gamma, beta, mean, rstd, dgamma, dbeta)
    rows, C = x2d.shape
    L = _lib.load()
    dx = torch.empty_like(x2d)
    ws = torch.empty(L.ssl4gie_bn_workspace_bytes(rows, C), dtype=torch.uint8, device=x2d.device)
    _lib.check(L.ssl4gie_bn_bwd_xmask(ptr(dy2d), ptr(x2d), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ptr(dx),
                                      ptr(dgamma), ptr(dbeta), int(accumulate), ptr(ws), code(x2d.dtype),
                                      rows, C, stream()), "bn_bwd_xmask")
    return dx


def maxpool3x3s2_fwd(x, coef=None, relu=False):
    """coef [2, C] (ops.bn_coef_partials): the pool runs over act(x coef[0] + coef[1]) — BatchNorm (+ ReLU) applied
    on the way in (ssl4gie_bn_maxpool3x3s2_fwd)"""
    B, H, W, C = _nhwc(x)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty(B, Ho, Wo, C, dtype=x.dtype, device=x.device)
    arg = torch.empty(B, Ho, Wo, C, dtype=torch.uint8, device=x.device)
    if coef is not None:
        _dev(coef); _f32(coef)
        assert coef.shape == (2, C) and coef.is_contiguous()
        _lib.check(_lib.load().ssl4gie_bn_maxpool3x3s2_fwd(ptr(x), ptr(coef), int(bool(relu)), ptr(y), ptr(arg),
                                                           code(x.dtype), B, H, W, C, stream()), "bn_maxpool_fwd")
        return y, arg
    _lib.check(_lib.load().ssl4gie_maxpool3x3s2_fwd(ptr(x), ptr(y), ptr(arg), code(x.dtype), B, H, W, C,
                                                    stream()), "maxpool_fwd")
    return y, arg


def maxpool3x3s2_bwd(dy, arg, H, W):
    B, Ho, Wo, C = _nhwc(dy)
    dx = torch.empty(B, H, W, C, dtype=dy.dtype, device=dy.device)
    _lib.check(_lib.load().ssl4gie_maxpool3x3s2_bwd(ptr(dy), ptr(arg), ptr(dx), code(dy.dtype), B, H, W,
                                                    C, stream()), "maxpool_bwd")
    return dx


def avgpool_fwd(x):
    B, H, W, C = _nhwc(x)
    y = torch.empty(B, C, dtype=torch.float32, device=x.device)
    _lib.check(_lib.load().ssl4gie_avgpool_fwd(ptr(x), ptr(y), code(x.dtype), B, H * W, C, stream()),
               "avgpool_fwd")
    return y


def avgpool_bwd(dy, H, W, dtype):
    _dev(dy)
    _f32(dy)
    B, C = dy.shape
    dx = torch.empty(B, H, W, C, dtype=dtype, device=dy.device)
    _lib.check(_lib.load().ssl4gie_avgpool_bwd(ptr(dy), ptr(dx), code(dtype), B, H * W, C, stream()),
               "avgpool_bwd")
    return dx


# ------------------------------------------------------------------ SyncBatchNorm pieces / MoCo EMA
def bn_stats(x2d, partials=None):
    """local batch statistics (mean, biased variance) of the rows of x2d, fp32 [C] each; from the
    producing GEMM's `partials` [parts, 2, C] when given"""
    _dev(x2d, partials)
    rows, C = x2d.shape
    L = _lib.load()
    mean = torch.empty(C, dtype=torch.float32, device=x2d.device)
    var = torch.empty(C, dtype=torch.float32, device=x2d.device)
    ws = torch.empty(L.ssl4gie_bn_workspace_bytes(rows, C), dtype=torch.uint8, device=x2d.device)
    if partials is not None:
        assert partials.dtype == torch.float32 and partials.shape[1:] == (2, C) and partials.is_contiguous()
        _lib.check(L.ssl4gie_bn_stats_partials(ptr(partials), partials.shape[0], ptr(mean), ptr(var),
                                               ptr(ws), rows, C, stream()), "bn_stats_partials")
        return mean, var
    _lib.check(L.ssl4gie_bn_stats(ptr(x2d), ptr(mean), ptr(var), ptr(ws), code(x2d.dtype), rows, C,
                                  stream()), "bn_stats")
    return mean, var


def bn_stats_from_partials(partials, rows):
    """local (mean, biased variance) of a map known only through its producer's partials [parts, 2, C] (the
    statistics-only product of linear_colstats_only: the map itself is never written)"""
    _dev(partials)
    C = partials.shape[2]
    assert partials.dtype == torch.float32 and partials.shape[1] == 2 and partials.is_contiguous()
    L = _lib.load()
    mean = torch.empty(C, dtype=torch.float32, device=partials.device)
    var = torch.empty(C, dtype=torch.float32, device=partials.device)
    ws = torch.empty(L.ssl4gie_bn_workspace_bytes(rows, C), dtype=torch.uint8, device=partials.device)
    _lib.check(L.ssl4gie_bn_stats_partials(ptr(partials), partials.shape[0], ptr(mean), ptr(var), ptr(ws), rows, C,
                                           stream()), "bn_stats_partials")
    return mean, var


def bn_bwd_reduce(dy2d, y2d, x2d, mean, rstd, relu, want_dres):
    _dev(dy2d, y2d, x2d, mean, rstd)
    rows, C = x2d.shape
    L = _lib.load()
    sums = torch.empty(2, C, dtype=torch.float32, device=x2d.device)
    dres = torch.empty_like(x2d) if want_dres else None
    ws = torch.empty(L.ssl4gie_bn_workspace_bytes(rows, C), dtype=torch.uint8, device=x2d.device)
    _lib.check(L.ssl4gie_bn_bwd_reduce(ptr(dy2d), ptr(y2d), ptr(x2d), ptr(mean), ptr(rstd), ptr(dres),
                                       ptr(sums), int(relu), ptr(ws), code(x2d.dtype), rows, C,
                                       stream()), "bn_bwd_reduce")
    return sums, dres


def bn_coef_stats(mean, rstd, gamma, beta):
    """coef [2, C] (y = x coef[0] + coef[1]) of a BatchNorm whose statistics are given — the GLOBAL ones a
    SyncBatchNorm exchange returned (ssl4gie_bn_coef_stats)"""
    _dev(mean, rstd, gamma, beta)
    C = mean.numel()
    coef = torch.empty(2, C, dtype=torch.float32, device=mean.device)
    _lib.check(_lib.load().ssl4gie_bn_coef_stats(ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), ptr(coef), C, stream()),
               "bn_coef_stats")
    return coef


def bn_apply_bits(x2d, coef, res):
    """relu(x coef[0] + coef[1] (+ res)) + the ReLU bit map (bf16): the apply half of bn_fwd_bits with given
    coefficients (ssl4gie_bn_apply_bits) -> (y, bits)"""
    _dev(x2d, coef, res)
    rows, C = x2d.shape
    assert x2d.dtype == torch.bfloat16 and coef.shape == (2, C) and coef.is_contiguous()
    y = torch.empty_like(x2d)
    bits = torch.empty(rows * C // 8, dtype=torch.uint8, device=x2d.device)
    _lib.check(_lib.load().ssl4gie_bn_apply_bits(ptr(x2d), ptr(coef), ptr(res), ptr(y), ptr(bits), code(x2d.dtype),
                                                 rows, C, stream()), "bn_apply_bits")
    return y, bits


def bn_bwd_reduce_bits(dy2d, bits, x2d, mean, rstd):
    """SyncBatchNorm + residual + ReLU backward, first half with the mask from the forward's bit map:
    -> (LOCAL sums [2, C], dres = the masked gradient) (ssl4gie_bn_bwd_reduce_bits)"""
    _dev(dy2d, bits, x2d, mean, rstd)
    rows, C = x2d.shape
    assert bits.dtype == torch.uint8 and bits.numel() == rows * C // 8
    L = _lib.load()
    sums = torch.empty(2, C, dtype=torch.float32, device=x2d.device)
    dres = torch.empty_like(x2d)
    ws = torch.empty(L.ssl4gie_bn_workspace_bytes(rows, C), dtype=torch.uint8, device=x2d.device)
    _lib.check(L.ssl4gie_bn_bwd_reduce_bits(ptr(dy2d), ptr(bits), ptr(x2d), ptr(mean), ptr(rstd), ptr(dres), ptr(sums),
                                            ptr(ws), code(x2d.dtype), rows, C, stream()), "bn_bwd_reduce_bits")
    return sums, dres


def bn_bwd_reduce_xmask(dy2d, x2d, gamma, beta, mean, rstd):
    """SyncBatchNorm + ReLU (no residual) backward, first half with the mask rebuilt from x: LOCAL sums [2, C]"""
    _dev(dy2d, x2d, gamma, beta, mean, rstd)
    rows, C = x2d.shape
    L = _lib.load()
    sums = torch.empty(2, C, dtype=torch.float32, device=x2d.device)
    ws = torch.empty(L.ssl4gie_bn_workspace_bytes(rows, C), dtype=torch.uint8, device=x2d.device)
    _lib.check(L.ssl4gie_bn_bwd_reduce_xmask(ptr(dy2d), ptr(x2d), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd),
                                             ptr(sums), ptr(ws), code(x2d.dtype), rows, C, stream()),
               "bn_bwd_reduce_xmask")
    return sums


def bn_bwd_apply_xmask(dy2d, x2d, gamma, beta, mean, rstd, sums, inv_count):
    """... second half: dx from the GLOBAL sums and 1 / (global row count)"""
    _dev(dy2d, x2d, gamma, beta, mean, rstd, sums)
    rows, C = x2d.shape
    dx = torch.empty_like(x2d)
    L = _lib.load()
    ws = torch.empty(L.ssl4gie_bn_workspace_bytes(rows, C), dtype=torch.uint8, device=x2d.device)
    _lib.check(L.ssl4gie_bn_bwd_apply_xmask(ptr(dy2d), ptr(x2d), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd),
                                            ptr(sums), float(inv_count), ptr(dx), ptr(ws), code(x2d.dtype), rows, C,
                                            stream()), "bn_bwd_apply_xmask")
    return dx


def bn_bwd_apply(dy2d, y2d, x2d, gamma, mean, rstd, sums, inv_count, relu):
    _dev(dy2d, y2d, x2d, gamma, mean, rstd, sums)
    rows, C = x2d.shape
    dx = torch.empty_like(x2d)
    L = _lib.load()
    ws = torch.empty(L.ssl4gie_bn_workspace_bytes(rows, C), dtype=torch.uint8, device=x2d.device)
    _lib.check(L.ssl4gie_bn_bwd_apply(ptr(dy2d), ptr(y2d), ptr(x2d), ptr(gamma), ptr(mean), ptr(rstd),
                                      ptr(sums), float(inv_count), ptr(dx), int(relu), ptr(ws),
                                      code(x2d.dtype), rows, C, stream()), "bn_bwd_apply")
    return dx


def ema_update(dst, src, m):
    """dst = dst * m + src * (1 - m) on flat fp32 tensors"""
    _dev(dst, src)
    _f32(dst, src)
    assert dst.numel() == src.numel()
    _lib.check(_lib.load().ssl4gie_ema_update(ptr(dst), ptr(src), float(m), dst.numel(), stream()),
               "ema_update")


# ------------------------------------------------------------------ detection pyramid glue (channels-last)
def maxpool2x2_fwd(x):
    B, H, W, C_ = _nhwc(x)
    y = torch.empty(B, H // 2, W // 2, C_, dtype=x.dtype, device=x.device)
    _lib.check(_lib.load().ssl4gie_maxpool2x2_fwd(ptr(x), ptr(y), code(x.dtype), B, H, W, C_, stream()),
               "maxpool2x2_fwd")
    return y


def maxpool2x2_bwd(x, dy):
    B, H, W, C_ = _nhwc(x)
    _dev(dy)
    dx = torch.empty_like(x)
    _lib.check(_lib.load().ssl4gie_maxpool2x2_bwd(ptr(x), ptr(dy), ptr(dx), code(x.dtype), B, H, W, C_,
                                                  stream()), "maxpool2x2_bwd")
    return dx


def gelu_map(x, dy=None):
    """gelu(x) (exact erf form), or dy * gelu'(x) when dy is given"""
    _dev(x, dy)
    out = torch.empty_like(x)
    _lib.check(_lib.load().ssl4gie_gelu_map(ptr(x), ptr(dy), ptr(out), code(x.dtype), x.numel(), stream()),
               "gelu_map")
    return out


def map_layernorm_fwd(x, w, bias, eps=1e-5):
    """nn.LayerNorm over everything but the batch axis; w / bias fp32 in x's element order"""
    _dev(x, w, bias)
    _f32(w)
    _f32(bias)
    B = x.shape[0]
    M = x.numel() // B
    assert w.numel() == M and bias.numel() == M and x.is_contiguous()
    L = _lib.load()
    y = torch.empty_like(x)
    mean = torch.empty(B, dtype=torch.float32, device=x.device)
    rstd = torch.empty(B, dtype=torch.float32, device=x.device)
    ws = torch.empty(L.ssl4gie_map_layernorm_workspace_bytes(B), dtype=torch.uint8, device=x.device)
    _lib.check(L.ssl4gie_map_layernorm_fwd(ptr(x), ptr(w), ptr(bias), ptr(y), ptr(mean), ptr(rstd), eps,
                                           ptr(ws), code(x.dtype), B, M, stream()), "map_layernorm_fwd")
    return y, mean, rstd


def map_layernorm_bwd(x, dy, w, mean, rstd, dw=None, db=None, accumulate=False):
    _dev(x, dy, w, mean, rstd, dw, db)
    B = x.shape[0]
    M = x.numel() // B
    L = _lib.load()
    dx = torch.empty_like(x)
    ws = torch.empty(L.ssl4gie_map_layernorm_workspace_bytes(B), dtype=torch.uint8, device=x.device)
    _lib.check(L.ssl4gie_map_layernorm_bwd(ptr(x), ptr(dy), ptr(w), ptr(mean), ptr(rstd), ptr(dx), ptr(dw),
                                           ptr(db), int(accumulate), ptr(ws), code(x.dtype), B, M,
                                           stream()), "map_layernorm_bwd")
    return dx


IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def normalize_u8(img_u8, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """uint8 HWC batch [B, H, W, 3] on the device -> fp32 NCHW, (x / 255 - mean) / std
    (ToTensor + Normalize of the reference's dataloaders, without the PIL / CPU round trip)"""
    _dev(img_u8)
    assert img_u8.dtype == torch.uint8 and img_u8.dim() == 4 and img_u8.shape[3] == 3 and img_u8.is_contiguous()
    B, H, W, _ = img_u8.shape
    out = torch.empty(B, 3, H, W, dtype=torch.float32, device=img_u8.device)
    m = (C.c_float * 3)(*mean)
    s = (C.c_float * 3)(*std)
    _lib.check(_lib.load().ssl4gie_normalize_u8(ptr(img_u8), ptr(out), m, s, B, H, W, stream()), "normalize_u8")
    return out
