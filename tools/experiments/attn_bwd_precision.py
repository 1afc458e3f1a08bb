#!/usr/bin/env python3
"""relative error of the bf16 attention backward against an fp64 evaluation, several seeds (A/B of two builds:
SSL4GIE_DEBUG_LIB=1 loads the other library)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ssl4gie_amd import ops

def ref(qkv, do, B, N, H, hd):
    t = qkv.double().requires_grad_(True)
    q, k, v = t.reshape(B, N, 3, H, hd).permute(2, 0, 3, 1, 4).unbind(0)
    s = (q @ k.transpose(-2, -1)) * hd ** -0.5
    o = (s.softmax(-1) @ v).transpose(1, 2).reshape(B, N, H * hd)
    o.backward(do.double())
    return t.grad

for hd, H in ((64, 12), (32, 16)):
    for N in (197, 50):
        errs = []
        for seed in range(6):
            g = torch.Generator().manual_seed(seed)
            B = 4
            qkv = (torch.randn(B, N, 3 * H * hd, generator=g) * 1.2).bfloat16()
            do = torch.randn(B, N, H * hd, generator=g).bfloat16()
            r = ref(qkv, do, B, N, H, hd)
            o, lse = ops.attn_fwd(qkv.cuda(), B, N, H, hd)
            d = ops.attn_bwd(qkv.cuda(), o, do.cuda(), lse, B, N, H, hd).double().cpu()
            D = H * hd
            e = [float((d.reshape(B, N, 3, D)[:, :, i] - r.reshape(B, N, 3, D)[:, :, i]).norm() / r.reshape(B, N, 3, D)[:, :, i].norm()) for i in range(3)]
            errs.append(e)
        m = torch.tensor(errs).mean(0)
        print(f"hd={hd} N={N}: rel err dq {m[0]:.4e} dk {m[1]:.4e} dv {m[2]:.4e}", flush=True)
