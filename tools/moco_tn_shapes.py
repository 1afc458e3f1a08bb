#!/usr/bin/env python3
"""which weight-gradient (TN) products of one MoCo-R50 step (B = 256) run on which kernel: logs every TN
descriptor that reaches ssl4gie_gemm (M, N, K, implicit-conv or not) and marks those the 256x256 TN kernel refuses
(K < 1024, K % 64 != 0 or M N < 65536: they fall to the round-1 128-tile kernel)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functools import partial
import torch
from ssl4gie_amd import ops, _lib
from ssl4gie_amd.Models.moco_v3.moco import builder
from ssl4gie_amd.Models.resnet import resnet50
_lib.load()
seen = collections.Counter()
raw = ops.gemm_raw
def spy(d, dev):
    if d.sAm == 1 and d.dtype_ab == _lib.BF16:  # TN layout: A is [K, M]
        big = d.K % 64 == 0 and d.K >= 1024 and d.M * d.N >= 65536
        seen[(d.M, d.N, d.K, bool(d.conv), "tn256" if (big or d.conv) else "OLD 128-tile")] += 1
    return raw(d, dev)
ops.gemm_raw = spy
torch.manual_seed(0)
B = int(os.environ.get("BENCH_B", "256"))
m = builder.MoCo_ResNet(partial(resnet50, zero_init_residual=True), 256, 4096, 1.0).cuda().set_precision("bf16")
g = torch.Generator("cpu").manual_seed(0)
x1 = torch.randn(B, 3, 224, 224, generator=g).cuda(); x2 = torch.randn(B, 3, 224, 224, generator=g).cuda()
loss = m(x1, x2, 0.99); loss.backward(); torch.cuda.synchronize()
for k, n in sorted(seen.items(), key=lambda kv: (kv[0][4], -kv[0][2])):
    print(f"{n:3d} x dW[{k[0]:5d},{k[1]:5d}] over K={k[2]:7d} conv={k[3]!s:5s} -> {k[4]}")
