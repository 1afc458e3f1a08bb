#!/usr/bin/env python3
"""Weight-gradient (TN) PAIR launches of the MAE ViT-B step (bs = 256) as the block executor issues them
(ssl4gie_gemm_tn_pair), timed two ways: `hot` = the same operands every iteration (they sit in the 256 MiB
Infinity Cache after the first pass: what tools/gemm_bench.py measures) and `cold` = operand sets rotated so that
every launch reads from HBM (what the training step sees: the operands were written a forward pass ago).
Prints time per launch, TFLOP/s and the time per 64-deep K-tile of one workgroup."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4gie_amd import ops, _lib

_lib.load()
dev = "cuda"
B = int(os.environ.get("BENCH_B", "256"))
enc_T, dec_T = B * 50, B * 197
# (name, T, (n_out_a, k_in_a), (n_out_b, k_in_b))
PAIRS = [("enc.fc2+fc1", enc_T, (768, 3072), (3072, 768)), ("enc.proj+qkv", enc_T, (768, 768), (2304, 768)),
         ("dec.fc2+fc1", dec_T, (512, 2048), (2048, 512)), ("dec.proj+qkv", dec_T, (512, 512), (1536, 512))]
ITERS = int(os.environ.get("TN_ITERS", "12"))
CUS = int(os.environ.get("SSL4GIE_COMPUTE_CUS", "240"))
FILL = int(os.environ.get("SSL4GIE_TN_FILL", "75"))


def splits_of(T, a, b):
    tiles = sum(((m + 255) // 256) * ((n + 255) // 256) for m, n in (a, b))
    s = (CUS * FILL // 100 + tiles // 2) // tiles
    s = max(1, min(s, (T // 64) // 8, 64))
    return tiles, s


def run(mode):
    tot = 0.0
    for name, T, a, b in PAIRS:
        per_set = 2 * T * (a[0] + a[1] + b[0] + b[1])
        nset = 1 if mode == "hot" else max(2, int(600e6 // per_set) + 1)
        sets = []
        for _ in range(nset):
            sets.append(tuple((torch.randn(T, c, device=dev) * 0.5).bfloat16() for c in (a[0], a[1], b[0], b[1])))
        ba = torch.empty(a[0], device=dev); bb = torch.empty(b[0], device=dev)
        def fn(i):
            dya, xa, dyb, xb = sets[i % nset]
            ops.linear_bwd_weight_pair(dya, xa, dyb, xb, ba, bb)
        for i in range(3): fn(i)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(ITERS): fn(i)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / ITERS * 1e3
        # the GEMM kernel alone: the library's launch profiler (HIP events around the tn256 launch)
        import ctypes as C
        L = _lib.load()
        _lib.check(L.ssl4gie_prof_begin(4 * ITERS + 8), "prof_begin")
        for i in range(ITERS): fn(i)
        ms = (C.c_double * _lib.PROF_KINDS)(); flp = (C.c_double * _lib.PROF_KINDS)(); nl = (C.c_longlong * _lib.PROF_KINDS)()
        _lib.check(L.ssl4gie_prof_collect(ms, flp, nl), "prof_collect")
        L.ssl4gie_prof_end()
        kus = ms[1] / max(1, nl[1]) * 1e3
        fl = 2.0 * T * (a[0] * a[1] + b[0] * b[1])
        tiles, sp = splits_of(T, a, b)
        kt = (T // 64) / sp
        tot += us
        print(f"TN-pair {mode:4s} {name:13s} T={T:6d} tiles={tiles:3d} splits={sp:2d} wgs={tiles * sp:3d} kt/wg={kt:6.1f} "
              f"{us:7.1f} us (GEMM + 2 slab reductions) | GEMM alone {kus:7.1f} us {fl / kus / 1e6:7.1f} TF/s  {kus / kt:5.2f} us per K-tile", flush=True)
        del sets
    print(f"TN-pair {mode} total {tot:.1f} us")


for mode in os.environ.get("TN_MODES", "hot,cold").split(","):
    run(mode)
