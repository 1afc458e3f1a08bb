#!/usr/bin/env python3
"""Fused-attention micro-benchmark on the shapes of the bench configs (bf16, random data)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4gie_amd import ops

torch.manual_seed(0)
CASES = [("mae.enc", 256, 50, 12, 64), ("mae.dec", 256, 197, 16, 32), ("vitb.full", 256, 197, 12, 64),
         ("det.window", 64, 256, 12, 64), ("det.global", 4, 4096, 12, 64)]
if os.environ.get("ATTN_SMALL"):  # the same heads at a batch whose tensors stay cache-resident: compute-bound time
    CASES = [("mae.dec/8", 32, 197, 16, 32), ("mae.dec", 256, 197, 16, 32), ("mae.dec/4", 64, 197, 16, 32),
             ("vitb/8", 32, 197, 12, 64), ("vitb.full", 256, 197, 12, 64)]


if os.environ.get("ATTN_CASES"):  # "name:B:N:H:hd,..."
    CASES = [(c.split(":")[0],) + tuple(int(v) for v in c.split(":")[1:]) for c in os.environ["ATTN_CASES"].split(",")]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for name, B, N, H, hd in CASES:
    qkv = torch.randn(B, N, 3 * H * hd, device="cuda").bfloat16()
    out, lse = ops.attn_fwd(qkv, B, N, H, hd)
    dout = torch.randn_like(out)
    tf = timeit(lambda: ops.attn_fwd(qkv, B, N, H, hd))
    tb = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, B, N, H, hd))
    ff = 4.0 * B * H * N * N * hd
    print(f"{name:10s} B={B} N={N} H={H} hd={hd}: fwd {tf*1e3:7.1f} us {ff/tf/1e9:6.1f} TF/s | "
          f"bwd {tb*1e3:7.1f} us {2.5*ff/tb/1e9:6.1f} TF/s", flush=True)
