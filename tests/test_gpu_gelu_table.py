"""The table form of the GELU epilogues of the 256 x 256 NT kernel (csrc/gemm256.h p_gelu_tab_read, csrc/gelu_table.h)
against its integer-exact emulation (tools/gen_gelu_table.py) and against the double-precision GELU.

Reference semantics: timm Mlp (fc1 -> GELU -> fc2) under autocast applies GELU to the half-precision fc1 output
(Models/mae/models_mae.py:39-41,53-55): Phi and gelu' at the bf16-rounded pre-activation.  The operands below are small
integers times powers of two, so the fp32 pre-activation u = acc + bias is exact and known on the host: the kernel's
outputs must equal the emulation BIT FOR BIT at the production shapes (which select the 256-wide tile)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import gen_gelu_table as gt  # noqa: E402


def test_gelu_table_header_is_current():
    """csrc/gelu_table.h holds exactly what the generator produces (a stale header would pass the GPU tests of an
    older table)"""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ssl4gie_amd", "csrc", "gelu_table.h")
    vals = []
    for line in open(path):
        line = line.strip()
        if line.startswith("0x"):
            vals += [int(v.rstrip("u"), 16) for v in line.rstrip(",").split(", ")]
    assert np.array_equal(np.array(vals, dtype=np.uint32), gt.table())


def test_gelu_table_accuracy_cpu():
    """the emulated outputs are bf16-accurate everywhere (output rounding + the argument-rounding term), including
    the clamped ends"""
    rs = np.random.RandomState(0)
    u = np.concatenate([rs.randn(400000) * 2.0, rs.randn(1000) * 1e-4, rs.randn(1000) * 40.0,
                        np.array([0.0, -0.0, 2.0 ** -12, -2.0 ** -12, 15.9375, 16.0, -16.0, 1e4, -1e4])]).astype(np.float32)
    d, g = gt.emulate(u)
    gf = (g.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    df = (d.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    t = torch.tensor(u, dtype=torch.float64, requires_grad=True)
    y = F.gelu(t)
    y.sum().backward()
    yr, dr = y.detach().numpy(), t.grad.numpy()
    # g = bf16(u Phi(bf16 u)): Phi's argument is off by <= 2^-8 |u| (first order: u^2 phi(u) 2^-8; 2 covers the
    # curvature in the tail), the output rounding by <= 2^-8 |g|; below Phi = 6e-5 (u < -3.85) the fp16 Phi is
    # subnormal (spacing 6e-8): absolute error <= 3e-8 |u| < 5e-7 there
    ud = u.astype(np.float64)
    phi = np.exp(-0.5 * ud * ud) / np.sqrt(2 * np.pi)
    bound_g = (2.0 * ud * ud * phi + 1.07 * np.abs(yr)) * 2.0 ** -8 + 5e-7  # 1.07: the fp16 Phi adds 2^-12
    assert np.all(np.abs(gf - yr) <= bound_g), float(np.max(np.abs(gf - yr) / bound_g))
    assert np.linalg.norm(gf - yr) / np.linalg.norm(yr) < 2.5e-3
    assert np.max(np.abs(df - dr)) < 8e-3 and np.linalg.norm(df - dr) / np.linalg.norm(dr) < 3e-3
    # sign and the trivial ends are exact
    assert np.array_equal(g[u > 16], gt.bf16_round_bits(u[u > 16])) and np.all(gf[u < -16] == 0)
    assert np.all(df[u > 16] == 1) and np.all(df[u < -16] == 0)


def _ints(shape, seed, lo, hi):
    return torch.randint(lo, hi, shape, generator=torch.Generator().manual_seed(seed)).float()


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(12800, 3072, 768), (50432, 2048, 512), (12744, 3064, 768)])
def test_gelu_pair_table_bit_exact(M, N, K):
    from ssl4gie_amd import _lib, ops
    dev = "cuda"
    x = _ints((M, K), 1, -2, 3)
    w = _ints((N, K), 2, -1, 2) * 2.0 ** -5
    w[5] = 0.0        # a column of tiny pre-activations (bias only)
    bias = _ints((N,), 3, -4096, 4097) * 2.0 ** -13
    bias[5] = 2.0 ** -13
    bias[6], bias[7], bias[9], bias[10] = 30.0, -30.0, 300.0, -300.0   # beyond the table on both sides
    u = (x.double() @ w.double().t() + bias.double()).float()          # exact: < 2^24 multiples of 2^-13
    assert torch.equal(u.double(), x.double() @ w.double().t() + bias.double())
    d_ref, g_ref = gt.emulate(u.numpy())
    xd, wd, bd = x.to(dev).bfloat16(), w.to(dev).bfloat16(), bias.to(dev)
    d1, g1 = ops.linear_fwd(xd, wd, bd, out_dtype=torch.bfloat16, epilogue=_lib.EPI_BIAS_GELU_GRAD)
    bits = lambda t: t.cpu().view(torch.int16).numpy().view(np.uint16).astype(np.uint32)
    assert np.array_equal(bits(g1), g_ref), "gelu"
    assert np.array_equal(bits(d1), d_ref), "gelu'"
    u2, g2 = ops.linear_fwd(xd, wd, bd, out_dtype=torch.bfloat16, epilogue=_lib.EPI_BIAS_GELU)
    assert np.array_equal(bits(g2), g_ref), "gelu (u, gelu) pair"
    assert np.array_equal(bits(u2), gt.bf16_round_bits(u.numpy())), "u"
    # and against the exact function
    t = u.double().requires_grad_(True)
    y = F.gelu(t)
    y.sum().backward()
    rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm())
    assert rel(g1.float(), y.detach()) < 3e-3 and rel(d1.float(), t.grad) < 3e-3
