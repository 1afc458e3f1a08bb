"""BatchNorm passes of the ResNet-50 trunk at the MoCo benchmark geometry (B = 256, 224 x 224, bf16 maps): time
and effective HBM rate of the forward (statistics + tail + apply; and apply alone from GEMM-epilogue partials) and
of the backward (reduce + tail + apply), per layer shape.  Bytes are the algorithmic ones (each tensor once).
python tools/bn_bench.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4gie_amd import ops, _lib

B = int(os.environ.get("BN_B", 256))
CASES = [  # name, rows, C, residual, relu
    ("stem bn1        112^2 x 64", B * 112 * 112, 64, False, True),
    ("layer1 bn1/2    56^2 x 64", B * 56 * 56, 64, False, True),
    ("layer1 bn3      56^2 x 256", B * 56 * 56, 256, True, True),
    ("layer1 ds       56^2 x 256", B * 56 * 56, 256, False, False),
    ("layer2.0 bn1    56^2 x 128", B * 56 * 56, 128, False, True),
    ("layer2 bn1/2    28^2 x 128", B * 28 * 28, 128, False, True),
    ("layer2 bn3      28^2 x 512", B * 28 * 28, 512, True, True),
    ("layer3 bn1/2    14^2 x 256", B * 14 * 14, 256, False, True),
    ("layer3 bn3      14^2 x 1024", B * 14 * 14, 1024, True, True),
    ("layer4 bn1/2    7^2 x 512", B * 7 * 7, 512, False, True),
    ("layer4 bn3      7^2 x 2048", B * 7 * 7, 2048, True, True),
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
    print(f"{'case':30s} {'fwd us':>8s} {'TB/s':>6s} | {'bwd us':>8s} {'TB/s':>6s}   (bytes: algorithmic, each tensor once)")
    tf = tb = tx = 0.0
    for name, rows, C, res, relu in CASES:
        x = torch.randn(rows, C, device="cuda").bfloat16()
        r = torch.randn(rows, C, device="cuda").bfloat16() if res else None
        g = torch.rand(C, device="cuda") + 0.5
        b = torch.randn(C, device="cuda") * 0.1
        rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
        dy = torch.randn(rows, C, device="cuda").bfloat16()
        dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        y, mean, rstd = ops.bn_fwd(x, g, b, r, rm, rv, 0.1, 1e-5, relu, True)
        n = rows * C * 2
        # forward: statistics read x; apply reads x (+ res), writes y
        fb = n * (3 + (1 if res else 0))
        # backward: reduce reads dy, x (+ y with ReLU) (+ writes dres); apply reads g, x (+ y if no dres), writes dx
        bb = n * ((2 + (1 if relu else 0) + (1 if res else 0)) + (2 + (1 if relu and not res else 0) + 1))
        t_f = timeit(lambda: ops.bn_fwd(x, g, b, r, rm, rv, 0.1, 1e-5, relu, True))
        t_b = timeit(lambda: ops.bn_bwd(dy, y if relu else None, x, g, mean, rstd, relu, res, dg, db, False))
        extra = ""
        if relu and not res:  # the mask rebuilt from x: 5 tensor passes instead of 7
            t_x = timeit(lambda: ops.bn_bwd_xmask(dy, x, g, b, mean, rstd, dg, db, False))
            extra = f" | xmask {t_x:8.1f} {n * 5 / t_x / 1e6:6.2f}"
            tx += t_x
        else:
            tx += t_b
        tf += t_f; tb += t_b
        print(f"{name:30s} {t_f:8.1f} {fb / t_f / 1e6:6.2f} | {t_b:8.1f} {bb / t_b / 1e6:6.2f}{extra}")
    print(f"sum: fwd {tf:.0f} us, bwd {tb:.0f} us, bwd with xmask where it applies {tx:.0f} us")


if __name__ == "__main__":
    main()
