#!/usr/bin/env python3
"""the weight-gradient products of MoCo-R50 that sat on the round-1 128-tile TN kernel: long contractions over
narrow outputs (layer1 1x1 convolutions) and short contractions over wide outputs (MLP heads); run with
SSL4GIE_TN256=0 (old kernel) / unset (routing) / 1 (256 kernels always)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4gie_amd import ops, _lib
_lib.load()
def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for M, N, T in ((128, 256, 802816), (256, 64, 802816), (64, 256, 802816), (64, 64, 802816),
                (256, 4096, 256), (4096, 256, 256), (4096, 2048, 256), (512, 128, 200704), (128, 512, 200704)):
    dy = torch.randn(T, M, device="cuda").bfloat16(); x = torch.randn(T, N, device="cuda").bfloat16()
    out = torch.empty(M, N, device="cuda")
    t = timeit(lambda: ops.linear_bwd_weight(dy, x, out=out))
    byt = 2.0 * T * (M + N)
    print(f"dW[{M:4d},{N:4d}] over {T:6d} rows: {t:7.1f} us  {byt / t / 1e6:5.2f} TB/s  {2.0 * T * M * N / t / 1e6:6.0f} TF/s", flush=True)
