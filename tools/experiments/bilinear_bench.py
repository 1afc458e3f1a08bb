#!/usr/bin/env python3
"""bilinear x2 forward / backward on the depth step's five maps (bs 128): time and HBM rate"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ssl4gie_amd import ops

def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

tf = tb = 0.0
for H, C in ((7, 256), (14, 256), (28, 256), (56, 256), (112, 128)):
    x = torch.randn(128, H, H, C, device="cuda").bfloat16()
    y = ops.bilinear2x_fwd(x)
    dy = torch.randn_like(y)
    a = timeit(lambda: ops.bilinear2x_fwd(x)); b = timeit(lambda: ops.bilinear2x_bwd(dy))
    mb = (x.numel() + y.numel()) * 2 / 1e6
    tf += a; tb += b
    print(f"{H:4d} -> {2*H:4d}  C={C}: fwd {a*1e3:7.1f} us {mb/a/1e3:5.2f} TB/s | bwd {b*1e3:7.1f} us {mb/b/1e3:5.2f} TB/s", flush=True)
print(f"sum: fwd {tf*1e3:.0f} us, bwd {tb*1e3:.0f} us")
