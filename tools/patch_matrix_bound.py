#!/usr/bin/env python3
"""What the last three patch-matrix geometries cost today, against an OPTIMISTIC bound for any patch-matrix-free
replacement (verdict r2-r4 item: stride-2 3x3 data gradient, weight gradient at Cin % 64 != 0).

stride-2 data gradient (torchvision Bottleneck.conv2 of layer{2,3,4}.0 at B = 256; DPT act_postprocess42.1 at
B = 128): today = dense product dcols[M_out, 9 Cin] = dy[M_out, Cout] W2[Cout, 9 Cin] + col2im scatter-add.  The
exact replacement is four parity-class convolutions with 1, 2, 2 and 4 taps (output pixels of one parity class
each); their gathered kernels cannot beat the PLAIN NT products of the same shapes ([M_out, Cin] outputs with
K = taps x Cout), which is what is timed here as the bound.
weight gradient at Cin = 96 (DPT layer1_rn, B = 128): today = patch matrix + TN product; bound = the gathered TN
weight gradient at Cin = 128 scaled by 96 / 128."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4gie_amd import ops, _lib
_lib.load()
BF = torch.bfloat16


def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


tot_now = tot_bound = 0.0
for name, B, H, W, Cin, Cout, per_step in (("layer2.0.conv2", 256, 56, 56, 128, 128, 2), ("layer3.0.conv2", 256, 28, 28, 256, 256, 2),
                                            ("layer4.0.conv2", 256, 14, 14, 512, 512, 2), ("act_postprocess42.1", 128, 14, 14, 768, 768, 1)):
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    M = B * Ho * Wo
    ld = ops.k_pad(9 * Cin, BF)
    dy = torch.randn(M, Cout, device="cuda").to(BF)
    w2 = (torch.randn(Cout, ld, device="cuda") * 0.02).to(BF)
    w2t = w2.t().contiguous()
    t_prod = timeit(lambda: ops.linear_bwd_data(dy, w2, w2t))
    dcols = ops.linear_bwd_data(dy, w2, w2t)
    t_c2i = timeit(lambda: ops.col2im3x3(dcols, B, H, W, Cin, 2))
    t_bound = 0.0
    for taps in (1, 2, 2, 4):   # one output parity class each: M_out pixels x Cin channels, K = taps * Cout
        a = torch.randn(M, taps * Cout, device="cuda").to(BF)
        wk = (torch.randn(Cin, taps * Cout, device="cuda") * 0.02).to(BF)
        t_bound += timeit(lambda: ops.linear_fwd(a, wk, None))
    print(f"{name:22s} B={B} {Cin}->{Cout} {H}x{W}: today {t_prod:7.1f} (product) + {t_c2i:6.1f} (col2im) = {t_prod + t_c2i:7.1f} us | "
          f"bound (4 plain class products) {t_bound:7.1f} us | gain <= {t_prod + t_c2i - t_bound:7.1f} us x {per_step}/step", flush=True)
    if per_step == 2:
        tot_now += 2 * (t_prod + t_c2i); tot_bound += 2 * t_bound
print(f"MoCo-R50 step (two views): today {tot_now:.0f} us, bound {tot_bound:.0f} us -> gain <= {tot_now - tot_bound:.0f} us of a ~55 000 us step "
      f"({100 * (tot_now - tot_bound) / 55000:.2f} %)")
# layer1_rn weight gradient, Cin = 96
B, H, W, Cin, Cout = 128, 56, 56, 96, 256
x = torch.randn(B, H, W, Cin, device="cuda").to(BF)
dy = torch.randn(B * H * W, Cout, device="cuda").to(BF)
ld = ops.k_pad(9 * Cin, BF)
def today():
    cols = ops.im2col3x3(x, 1, False, ld)
    return ops.linear_bwd_weight(dy, cols)
t_now = timeit(today)
x128 = torch.randn(B, H, W, 128, device="cuda").to(BF)
t128 = timeit(lambda: ops.conv3x3_bwd_weight(dy, x128, 1, False))
print(f"layer1_rn weight gradient 96->256 @56 B=128: today (patch matrix + TN) {t_now:7.1f} us | gathered TN at Cin = 128: {t128:7.1f} us, "
      f"x 96/128 = {0.75 * t128:7.1f} us | gain <= {t_now - 0.75 * t128:7.1f} us of a ~38 500 us depth step ({100 * (t_now - 0.75 * t128) / 38500:.2f} %)")
