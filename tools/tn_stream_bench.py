import os, sys
sys.path.insert(0, "/root/repo")
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
T = 802816
for M, N in ((256, 256), (128, 256), (64, 256), (256, 64), (64, 64), (256, 512), (512, 512)):
    dy = torch.randn(T, M, device="cuda").bfloat16(); x = torch.randn(T, N, device="cuda").bfloat16()
    t = timeit(lambda: ops.linear_bwd_weight(dy, x))
    byt = 2.0 * T * (M + N)
    print(f"dW[{M:4d},{N:4d}] over {T} rows: {t:7.1f} us  {byt / t / 1e6:5.2f} TB/s  {2.0 * T * M * N / t / 1e6:6.0f} TF/s")
