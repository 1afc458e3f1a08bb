#!/usr/bin/env python3
"""Probe: does running the MAE step as TWO half-batches on two streams fill the chip better than one
batch on one stream?  Forward only (no gradient race to handle), then forward+backward with two
independent model replicas (upper bound of what interleaving could give)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ssl4gie_amd.Models.mae import models_mae

dev = "cuda"
torch.manual_seed(0)
B = 256
m = models_mae.mae_vit_base_patch16(norm_pix_loss=True).to(dev).set_precision("bf16")
imgs = torch.randn(B, 3, 224, 224, device=dev)

def timeit(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3

def fwd_one():
    with torch.no_grad():
        m(imgs, mask_ratio=0.75)

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
halves = [imgs[:B // 2].contiguous(), imgs[B // 2:].contiguous()]
def fwd_two():
    cur = torch.cuda.current_stream()
    for s, x in zip((s1, s2), halves):
        s.wait_stream(cur)
        with torch.cuda.stream(s), torch.no_grad():
            m(x, mask_ratio=0.75)
    cur.wait_stream(s1); cur.wait_stream(s2)

def fwd_two_serial():
    with torch.no_grad():
        for x in halves:
            m(x, mask_ratio=0.75)

print("fwd  one batch      %.3f ms" % timeit(fwd_one))
print("fwd  two halves, 1 stream %.3f ms" % timeit(fwd_two_serial))
print("fwd  two halves, 2 streams %.3f ms" % timeit(fwd_two))

# forward + backward: two replicas (separate arenas: no gradient race) on two streams
import copy
m2 = models_mae.mae_vit_base_patch16(norm_pix_loss=True).to(dev).set_precision("bf16")
def fb_one():
    for p in m.parameters(): p.grad = None
    loss, _, _ = m(imgs, mask_ratio=0.75)
    loss.backward()
def fb_two():
    cur = torch.cuda.current_stream()
    losses = []
    for mm, s, x in ((m, s1, halves[0]), (m2, s2, halves[1])):
        for p in mm.parameters(): p.grad = None
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            l, _, _ = mm(x, mask_ratio=0.75)
            losses.append(l)
    for l, s in zip(losses, (s1, s2)):
        with torch.cuda.stream(s):
            l.backward()
    cur.wait_stream(s1); cur.wait_stream(s2)
def fb_two_serial():
    for mm, x in ((m, halves[0]), (m2, halves[1])):
        for p in mm.parameters(): p.grad = None
        l, _, _ = mm(x, mask_ratio=0.75)
        l.backward()
print("f+b  one batch      %.3f ms" % timeit(fb_one))
print("f+b  two halves (2 replicas), 1 stream %.3f ms" % timeit(fb_two_serial))
print("f+b  two halves (2 replicas), 2 streams %.3f ms" % timeit(fb_two))
