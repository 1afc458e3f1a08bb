#!/usr/bin/env python3
"""In-kernel timeline of gemm_nt256 per output tile (SSL4GIE_NT256_NOEPI=4 stamps): K-loop, epilogue issue,
and how long the NEXT tile's first / second K-tile take (its counted vmcnt wait sits behind the stores)."""
import ctypes, os, sys
os.environ.setdefault("SSL4GIE_NT256_NOEPI", "4")  # "8": wave 4's view (a wr = 1 wave)
os.environ["SSL4GIE_DEBUG_LIB"] = "1"  # the stamps exist in the debug library only
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ssl4gie_amd import ops, _lib

L = _lib.load()
dev = "cuda"
B = 256
shapes = [("dec.qkv", B * 197, 1536, 512, "bias"), ("dec.fc1", B * 197, 2048, 512, "gelu"), ("dec.proj", B * 197, 512, 512, "res"),
          ("dec.dqkv", B * 197, 512, 1536, "none"), ("enc.fc1", B * 50, 3072, 768, "gelu"), ("enc.qkv", B * 50, 2304, 768, "bias")]
prev = np.zeros((16, 16, 16), dtype=np.uint64)
for name, T, n, k, ep in shapes:
    x = (torch.randn(T, k, device=dev) * 0.5).bfloat16()
    w = (torch.randn(n, k, device=dev) * 0.05).bfloat16()
    b = torch.randn(n, device=dev)
    res = torch.randn(T, n, device=dev) if ep == "res" else None
    if ep == "bias": fn = lambda: ops.linear_fwd(x, w, b)
    elif ep == "res": fn = lambda: ops.linear_fwd(x, w, b, out_dtype=torch.float32, epilogue=_lib.EPI_BIAS_RESIDUAL, residual=res)
    elif ep == "gelu": fn = lambda: ops.linear_fwd(x, w, b, epilogue=_lib.EPI_BIAS_GELU_GRAD)
    else: fn = lambda: ops.linear_fwd(x, w, None)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    buf = np.zeros((16, 16, 16), dtype=np.uint64)
    _lib.check(L.ssl4gie_debug_nt256_stamps(buf.ctypes.data, buf.nbytes), "stamps")
    nk = k // 64
    rows, blocks = [], []
    for wg in range(16):
        for ti in range(0, 15):
            s = buf[wg, ti].astype(np.int64)
            if s[0] == 0 or s[4] <= s[0] or s[3] <= s[2] or (buf[wg, ti] == prev[wg, ti]).all():
                continue  # empty, incomplete, or left over from an earlier (longer) launch
            rows.append(((s[1] - s[0]) / 100.0, (s[2] - s[1]) / 100.0, (s[3] - s[2]) / 100.0, (s[4] - s[3]) / 100.0))
            if s[12] > s[1]:  # per-row-block stamps of the bf16 epilogues (debug library)
                blocks.append([(s[5 + k] - (s[1] if k == 0 else s[4 + k])) / 100.0 for k in range(8)])
    prev = buf.copy()
    if not rows:
        print(name, "no stamps"); continue
    r = np.median(np.array(rows), axis=0)
    print(f"{name:9s} nk={nk:2d}: K-loop {r[0]:6.2f} us ({r[0]/nk:5.2f}/K-tile) | epilogue issue {r[1]:6.2f} us | "
          f"next tile: 1st K-tile {r[2]:6.2f} us, 2nd K-tile {r[3]:6.2f} us   [{len(rows)} tiles]"
          + ("" if not blocks else "  row blocks: " + " ".join(f"{v:.2f}" for v in np.median(np.array(blocks), axis=0))))
