"""ResNet stem (7x7 stride-2 conv, 3 -> 64) at 224 x 224: patch-matrix path (im2col + GEMMs) against the
direct kernels (packed image).  python tools/stem_bench.py [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4gie_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
BF = torch.bfloat16


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


x = torch.randn(B, 3, 224, 224, device="cuda")
w = torch.randn(64, 3, 7, 7, device="cuda") * 0.05
dy = torch.randn(B, 112, 112, 64, device="cuda").to(BF)
cols, Ho, Wo = ops.stem_im2col7x7(x, BF)
ld = cols.shape[1]
w2 = torch.zeros(64, ld, device="cuda", dtype=BF)
w2[:, :147] = w.permute(0, 2, 3, 1).reshape(64, 147).to(BF)
t_im = timeit(lambda: ops.stem_im2col7x7(x, BF))
t_f = timeit(lambda: ops.linear_fwd(cols, w2, None, out_dtype=BF, colstats=True))
t_w = timeit(lambda: ops.linear_bwd_weight(dy.view(-1, 64), cols))
print(f"patch matrix: im2col {t_im:7.1f} us | fwd GEMM (+stats) {t_f:7.1f} us | wgrad GEMM {t_w:7.1f} us")
packed = ops.stem7x7_pack(x)
w2s = ops.stem7x7_weight(w).to(BF)
t_p = timeit(lambda: ops.stem7x7_pack(x))
t_df = timeit(lambda: ops.stem7x7_fwd(packed, w2s, B, 224, 224, colstats=True))
t_dw = timeit(lambda: ops.stem7x7_wgrad(dy, packed, B, 224, 224))
print(f"direct:       pack   {t_p:7.1f} us | fwd (+stats)      {t_df:7.1f} us | wgrad      {t_dw:7.1f} us")
y0, _ = ops.linear_fwd(cols, w2, None, out_dtype=BF, colstats=True)
y1, _ = ops.stem7x7_fwd(packed, w2s, B, 224, 224, colstats=True)
print("fwd rel diff", float((y1.float().view(-1, 64) - y0.float()).norm() / y0.float().norm()))
