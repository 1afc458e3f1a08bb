#!/usr/bin/env python3
"""Per-shape GEMM micro-benchmark on the shapes of the MAE ViT-B step (bs=256): TFLOP/s of the NT
(forward / data-gradient) and TN (weight-gradient) bf16 paths, random operands."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4gie_amd import ops, _lib

dev = "cuda"
torch.manual_seed(0)
B = int(os.environ.get("BENCH_B", "256"))
enc_T, dec_T = B * 50, B * 197
NT = [  # (name, T, n_out, k_in, epilogue)
    ("enc.qkv", enc_T, 2304, 768, "bias"), ("enc.proj", enc_T, 768, 768, "res"),
    ("enc.fc1", enc_T, 3072, 768, "gelu"), ("enc.fc2", enc_T, 768, 3072, "res"),
    ("enc.dfc2", enc_T, 3072, 768, "dgelu"), ("enc.dqkv", enc_T, 768, 2304, "none"),
    ("dec.qkv", dec_T, 1536, 512, "bias"), ("dec.proj", dec_T, 512, 512, "res"),
    ("dec.fc1", dec_T, 2048, 512, "gelu"), ("dec.fc2", dec_T, 512, 2048, "res"),
    ("dec.dfc2", dec_T, 2048, 512, "dgelu"), ("dec.dqkv", dec_T, 512, 1536, "none"),
    ("dec.pred", dec_T, 768, 512, "biasf32"),
]
TN = [("enc.dWqkv", 2304, 768, enc_T), ("enc.dWproj", 768, 768, enc_T), ("enc.dWfc1", 3072, 768, enc_T),
      ("enc.dWfc2", 768, 3072, enc_T), ("dec.dWqkv", 1536, 512, dec_T), ("dec.dWproj", 512, 512, dec_T),
      ("dec.dWfc1", 2048, 512, dec_T), ("dec.dWfc2", 512, 2048, dec_T)]

def timeit(fn, iters=int(os.environ.get("GEMM_ITERS", "20"))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

if os.environ.get("PROBE"):
    NT = [("cube4k", 4096, 4096, 4096, "none"), ("cube8k", 8192, 8192, 8192, "none"),
          ("m2k_k8k", 2048, 2048, 8192, "none"), ("tall", 50432, 512, 512, "none"),
          ("tall_k2k", 50432, 512, 2048, "none"), ("wide", 12800, 3072, 768, "none"),
          ("wide_k3k", 12800, 3072, 3072, "none")]
    TN = []
if os.environ.get("GEMM_CASES"):  # comma-separated names
    keep = set(os.environ["GEMM_CASES"].split(","))
    NT = [c for c in NT if c[0] in keep]; TN = [c for c in TN if c[0] in keep]
tot_ms = tot_fl = 0
for name, T, n, k, ep in NT:
    x = (torch.randn(T, k, device=dev) * 0.5).bfloat16()
    w = (torch.randn(n, k, device=dev) * 0.05).bfloat16()
    b = torch.randn(n, device=dev)
    res = torch.randn(T, n, device=dev) if ep == "res" else None
    aux = torch.randn(T, n, device=dev).bfloat16() if ep == "dgelu" else None
    if ep == "bias": fn = lambda: ops.linear_fwd(x, w, b)
    elif ep == "biasf32": fn = lambda: ops.linear_fwd(x, w, b, out_dtype=torch.float32)
    elif ep == "res": fn = lambda: ops.linear_fwd(x, w, b, out_dtype=torch.float32, epilogue=_lib.EPI_BIAS_RESIDUAL, residual=res)
    elif ep == "gelu": fn = lambda: ops.linear_fwd(x, w, b, epilogue=_lib.EPI_BIAS_GELU_GRAD)
    elif ep == "dgelu": fn = lambda: ops.linear_bwd_data(x, w.t().contiguous(), w, dgelu_aux=aux) if False else ops.linear_fwd(x, w, None)
    else: fn = lambda: ops.linear_fwd(x, w, None)
    if ep == "dgelu":
        d = ops._desc(T, n, k, _lib.BF16, _lib.BF16)
        y = torch.empty(T, n, dtype=torch.bfloat16, device=dev)
        d.A, d.sAm, d.sAk = x.data_ptr(), k, 1
        d.B, d.sBk, d.sBn = w.data_ptr(), 1, k
        d.C, d.ldc = y.data_ptr(), n
        d.epilogue, d.aux = _lib.EPI_MUL_AUX, aux.data_ptr()
        fn = lambda: ops.gemm_raw(d, dev)
    ms = timeit(fn)
    fl = 2.0 * T * n * k
    tot_ms += ms; tot_fl += fl
    print(f"NT {name:10s} M={T:6d} N={n:5d} K={k:5d} {ep:8s} {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s", flush=True)
print(f"NT total {tot_ms:.3f} ms  {tot_fl/tot_ms/1e9:.1f} TF/s")
if os.environ.get("GEMM_SKIP_TN"): TN = []
tot_ms = tot_fl = 0
for name, n, k, T in TN:
    dy = (torch.randn(T, n, device=dev) * 0.5).bfloat16()
    x = (torch.randn(T, k, device=dev) * 0.5).bfloat16()
    out = torch.empty(n, k, device=dev)
    ms = timeit(lambda: ops.linear_bwd_weight(dy, x, out=out))
    fl = 2.0 * T * n * k
    tot_ms += ms; tot_fl += fl
    print(f"TN {name:10s} M={n:5d} N={k:5d} K={T:6d} {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF/s", flush=True)
if TN: print(f"TN total {tot_ms:.3f} ms  {tot_fl/tot_ms/1e9:.1f} TF/s")
