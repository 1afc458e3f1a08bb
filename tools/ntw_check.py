#!/usr/bin/env python3
"""gemm_nt256w (four-wave NT kernel with the trickled epilogue) against gemm_nt256 on the same operands:
every epilogue kind, MAE shapes and ragged ones.  Both are bf16-output kernels of the same products, so
they agree to bf16 rounding of the branch output (the w kernel rounds acc + bias to bf16 BEFORE the
residual add / GELU / aux multiply: the reference's autocast data flow)."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def run(mode):
    import torch
    from ssl4gie_amd import ops, _lib
    dev = "cuda"
    out = {}
    shapes = [(12800, 768, 768), (50432, 512, 512), (2560, 384, 512), (1000, 264, 1024), (300, 3072, 768), (4096, 2048, 512)]
    for (T, n, k) in shapes:
        g = torch.Generator().manual_seed(T + n + k)
        x = (torch.randn(T, k, generator=g) * 0.5).to(dev).bfloat16()
        w = (torch.randn(n, k, generator=g) * 0.05).to(dev).bfloat16()
        b = torch.randn(n, generator=g).to(dev)
        res = torch.randn(T, n, generator=g).to(dev)
        aux = torch.randn(T, n, generator=g).to(dev).bfloat16()
        for it in range(2):  # twice: the second launch runs with warm caches / different timing
            y_bias = ops.linear_fwd(x, w, b)
            y_none = ops.linear_fwd(x, w, None)
            y_res = ops.linear_fwd(x, w, b, out_dtype=torch.float32, epilogue=_lib.EPI_BIAS_RESIDUAL, residual=res)
            y_gelu = ops.linear_fwd(x, w, b, epilogue=_lib.EPI_BIAS_GELU_GRAD)
            d = ops._desc(T, n, k, _lib.BF16, _lib.BF16)
            y_aux = torch.empty(T, n, dtype=torch.bfloat16, device=dev)
            d.A, d.sAm, d.sAk = x.data_ptr(), k, 1
            d.B, d.sBk, d.sBn = w.data_ptr(), 1, k
            d.C, d.ldc = y_aux.data_ptr(), n
            d.epilogue, d.aux = _lib.EPI_MUL_AUX, aux.data_ptr()
            ops.gemm_raw(d, dev)
        torch.cuda.synchronize()
        ref = x.double() @ w.double().t()
        rb = ref + b.double()
        def err(a, r):
            return float((a.double() - r).abs().max() / r.abs().max())
        gg = y_gelu if isinstance(y_gelu, (tuple, list)) else (y_gelu,)
        gel = torch.nn.functional.gelu(rb)
        out[f"{T}x{n}x{k}"] = dict(bias=err(y_bias, rb), none=err(y_none, ref), res=err(y_res, rb + res.double()),
                                   aux=err(y_aux, ref * aux.double()),
                                   gelu=[err(t, gel) for t in gg], nan=bool(torch.isnan(y_res).any() or torch.isnan(y_bias.float()).any()))
    print(json.dumps(out))

if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for mode in ("0", "1"):
            env = dict(os.environ, SSL4GIE_NT256W=mode)
            r = subprocess.run([sys.executable, __file__, mode], env=env, capture_output=True, text=True)
            print("NT256W=" + mode, r.stdout.strip()[-3000:], r.stderr.strip()[-1500:] if r.returncode else "")
