"""The 1x1 convolutions of ResNet-50 at the MoCo benchmark geometry (B = 256, 224 x 224, bf16) as the GEMMs the
engine runs: forward with BatchNorm statistics in the epilogue (colstats), the statistics-only product, the
BatchNorm-fused product (EPI_AFFINE_AUX_RELU), and the data gradient; time and effective HBM rate (algorithmic
bytes: each operand once).  python tools/r50_gemm_bench.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4gie_amd import ops, _lib

B = int(os.environ.get("R50_B", 256))
CASES = [  # name, rows, K, N
    ("layer1.0 conv1   64>64   @56", B * 3136, 64, 64),
    ("layer1.x conv1  256>64   @56", B * 3136, 256, 64),
    ("layer1 conv3     64>256  @56", B * 3136, 64, 256),
    ("layer2.0 conv1  256>128  @56", B * 3136, 256, 128),
    ("layer2.x conv1  512>128  @28", B * 784, 512, 128),
    ("layer2 conv3    128>512  @28", B * 784, 128, 512),
    ("layer2.0 ds     256>512  @28", B * 784, 256, 512),
    ("layer3.x conv1 1024>256  @14", B * 196, 1024, 256),
    ("layer3 conv3    256>1024 @14", B * 196, 256, 1024),
    ("layer4.x conv1 2048>512  @7", B * 49, 2048, 512),
    ("layer4 conv3    512>2048 @7", B * 49, 512, 2048),
]


def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3  # us


def main():
    _lib.load()
    print(f"{'case':32s} {'fwd':>8s} | {'fwd+stats':>10s} {'TB/s':>5s} {'TF/s':>5s} | {'stats only':>10s} | {'affine':>8s} {'TB/s':>5s} | {'dgrad':>8s} {'TB/s':>5s} | {'wgrad':>8s} {'TB/s':>5s}")
    tot = [0.0] * 5
    for name, T, K, N in CASES:
        x = torch.randn(T, K, device="cuda").bfloat16()
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
        wt = w.t().contiguous()
        dy = torch.randn(T, N, device="cuda").bfloat16()
        aux = torch.randn(T, N, device="cuda").bfloat16()
        sc, sh = torch.rand(N, device="cuda") + 0.5, torch.randn(N, device="cuda")
        byt = 2.0 * T * (K + N)
        tp = timeit(lambda: ops.linear_fwd(x, w, None))
        t0 = timeit(lambda: ops.linear_fwd(x, w, None, colstats=True))
        t1 = timeit(lambda: ops.linear_colstats_only(x, w))
        t2 = timeit(lambda: ops.linear_affine_fwd(x, w, sc, sh, aux, True))
        t3 = timeit(lambda: ops.linear_bwd_data(dy, w, wt))
        t4 = timeit(lambda: ops.linear_bwd_weight(dy, x))
        for i, t in enumerate((t0, t1, t2, t3, t4)):
            tot[i] += t
        print(f"{name:32s} {tp:8.1f} | {t0:10.1f} {byt / t0 / 1e6:5.2f} {2.0 * T * K * N / t0 / 1e6:5.0f} | {t1:10.1f} | "
              f"{t2:8.1f} {(byt + 2.0 * T * N) / t2 / 1e6:5.2f} | {t3:8.1f} {byt / t3 / 1e6:5.2f} | {t4:8.1f} {byt / t4 / 1e6:5.2f}")
    print("sum: " + " ".join(f"{t:.0f}" for t in tot) + " us")


if __name__ == "__main__":
    main()
