#!/usr/bin/env python3
"""ResNet-50 stem max-pool (forward with bn + relu on the way in, backward) and the stride-2 col2im at bs 256"""
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
    return s.elapsed_time(e) / iters * 1e3

B = 256
x = torch.randn(B, 112, 112, 64, device="cuda").bfloat16()
coef = torch.stack([torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.1]).contiguous()
y, arg = ops.maxpool3x3s2_fwd(x, coef, True)
dy = torch.randn_like(y)
print(f"maxpool fwd (bn+relu in) {timeit(lambda: ops.maxpool3x3s2_fwd(x, coef, True)):7.1f} us | bwd {timeit(lambda: ops.maxpool3x3s2_bwd(dy, arg, 112, 112)):7.1f} us", flush=True)
for H, C in ((56, 128), (28, 256), (14, 512)):
    Ho = H // 2
    dcols = torch.randn(B * Ho * Ho, 9 * C, device="cuda").bfloat16()
    print(f"col2im3x3 s2 {H:3d} C={C}: {timeit(lambda: ops.col2im3x3(dcols, B, H, H, C, 2)):7.1f} us", flush=True)
