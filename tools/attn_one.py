#!/usr/bin/env python3
"""A few launches of the fused attention kernels on ONE shape (ATTN_CASE = B:N:H:hd, default the MAE decoder's) —
the subject for the counter passes of tools/gpu_pmc_sq.sh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4gie_amd import ops
B, N, H, hd = (int(v) for v in os.environ.get("ATTN_CASE", "256:197:16:32").split(":"))
torch.manual_seed(0)
qkv = torch.randn(B, N, 3 * H * hd, device="cuda").bfloat16()
out, lse = ops.attn_fwd(qkv, B, N, H, hd)
dout = torch.randn_like(out)
for _ in range(int(os.environ.get("ATTN_ITERS", "4"))):
    ops.attn_fwd(qkv, B, N, H, hd)
    ops.attn_bwd(qkv, out, dout, lse, B, N, H, hd)
torch.cuda.synchronize()
