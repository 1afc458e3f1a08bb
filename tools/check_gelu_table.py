#!/usr/bin/env python3
"""where (if anywhere) the table epilogue differs from its emulation: counts and samples per output"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import gen_gelu_table as gt
from ssl4gie_amd import _lib, ops
M, N, K = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (12800, 3072, 768)))
g = lambda s: torch.Generator().manual_seed(s)
x = torch.randint(-2, 3, (M, K), generator=g(1)).float()
w = torch.randint(-1, 2, (N, K), generator=g(2)).float() * 2.0 ** -5
bias = torch.randint(-4096, 4097, (N,), generator=g(3)).float() * 2.0 ** -13
bias[6], bias[7], bias[9], bias[10] = 30.0, -30.0, 300.0, -300.0
u = (x.double() @ w.double().t() + bias.double()).float().numpy()
d_ref, g_ref = gt.emulate(u)
d1, g1 = ops.linear_fwd(x.cuda().bfloat16(), w.cuda().bfloat16(), bias.cuda(), out_dtype=torch.bfloat16, epilogue=_lib.EPI_BIAS_GELU_GRAD)
bits = lambda t: t.cpu().view(torch.int16).numpy().view(np.uint16).astype(np.uint32)
for name, got, ref in (("gelu", bits(g1), g_ref), ("dgelu", bits(d1), d_ref)):
    bad = np.argwhere(got != ref)
    print(name, "mismatches", len(bad), "of", got.size)
    for r, c in bad[:12]:
        print("  row %d col %d u=%r got=%#06x ref=%#06x  idx=%d" % (r, c, float(u[r, c]), got[r, c], ref[r, c], gt.lookup_index(u[r:r+1, c:c+1])[0, 0]))
    if len(bad):
        print("  rows mod 256:", np.unique(bad[:, 0] % 256)[:40], " cols mod 256:", np.unique(bad[:, 1] % 256)[:40])
        diff = np.abs(got[bad[:, 0], bad[:, 1]].astype(np.int64) - ref[bad[:, 0], bad[:, 1]].astype(np.int64))
        print("  |bit diff| histogram:", np.bincount(np.minimum(diff, 8)))
