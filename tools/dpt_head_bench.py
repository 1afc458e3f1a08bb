"""DPT output head (DPT_decoder.py:469-481) convolution timings at the depth-finetune bench batch:
forward, data gradient and weight gradient of output_conv.0 (256->128 @112^2) and output_conv.2
(128->32 @224^2) on the paths dpt_engine.Conv3x3Fn takes.  python tools/dpt_head_bench.py [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4gie_amd import ops, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
CASES = [("output_conv.0 256>128 @112", 112, 256, 128), ("output_conv.2 128>32 @224", 224, 128, 32),
         ("refinenet rcu 256>256 @56", 56, 256, 256)]


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, HW, Ci, Co in CASES:
    x = torch.randn(B, HW, HW, Ci, device="cuda").bfloat16()
    w2 = (torch.randn(Co, 9 * Ci, device="cuda") * 0.02).bfloat16()
    wd = (torch.randn(Ci, 9 * Co, device="cuda") * 0.02).bfloat16()
    bias = torch.zeros(Co, device="cuda")
    dy = torch.randn(B, HW, HW, Co, device="cuda").bfloat16()
    fl = 2.0 * B * HW * HW * Co * 9 * Ci
    f = timeit(lambda: ops.conv3x3_fwd(x, w2, bias, 1, True))
    if ops.conv3x3_implicit_ok(dy, 1, Ci):
        d = timeit(lambda: ops.conv3x3_fwd(dy, wd, None, 1, False, relu_mask=x))
        how = "implicit"
    else:
        ld2 = ops.k_pad(9 * Co, torch.bfloat16)
        wdp = torch.zeros(Ci, ld2, device="cuda", dtype=torch.bfloat16)
        d = timeit(lambda: ops.relu_bwd(x, ops.linear_fwd(ops.im2col3x3(dy, 1, False, ld2), wdp, None,
                                                        out_dtype=torch.bfloat16).view(B, HW, HW, Ci)))
        how = "im2col"
    w = timeit(lambda: ops.conv3x3_bwd_weight(dy.view(-1, Co), x, 1, True, bias_out=bias))
    if os.environ.get('FORCE_DIRECT'):
        fd = timeit(lambda: ops.conv3x3_direct_fwd(x, w2, bias, relu=True))
        print(f'{name:30s} B={B}: DIRECT(forced) fwd {fd:8.1f} us ({fl / fd / 1e6:6.1f} TF/s)', flush=True)
    if os.environ.get('FORCE_DIRECT') and _lib.load().ssl4gie_conv3x3_direct_wgrad_ok(B, HW, HW, Ci, Co):
        wf = timeit(lambda: ops.conv3x3_direct_wgrad(dy, x, relu=True, bias_out=bias))
        print(f'{name:30s} B={B}: DIRECT(forced) wgrad {wf:8.1f} us ({fl / wf / 1e6:6.1f} TF/s)', flush=True)
    if ops.conv3x3_direct_ok(x, Co):
        fd = timeit(lambda: ops.conv3x3_direct_fwd(x, w2, bias, relu=True))
        dd = timeit(lambda: ops.conv3x3_direct_fwd(dy, wd, None, relu_mask=x))
        wdd = timeit(lambda: ops.conv3x3_direct_wgrad(dy, x, relu=True, bias_out=bias)) if ops.conv3x3_direct_wgrad_ok(x, Co) else float('nan')
        cs = timeit(lambda: ops.colsum(dy.view(-1, Co), out=bias))
        print(f"{name:30s} B={B}: DIRECT fwd {fd:8.1f} us ({fl / fd / 1e6:6.1f} TF/s) | dgrad {dd:8.1f} us "
              f"({fl / dd / 1e6:6.1f}) | wgrad {wdd:8.1f} us ({fl / wdd / 1e6:6.1f}) + colsum {cs:7.1f} us", flush=True)
    print(f"{name:30s} B={B}: fwd {f:8.1f} us ({fl / f / 1e6:6.1f} TF/s) | dgrad[{how}] {d:8.1f} us "
          f"({fl / d / 1e6:6.1f}) | wgrad {w:8.1f} us ({fl / w / 1e6:6.1f})", flush=True)
    del x, dy
    torch.cuda.empty_cache()
