"""Host-side enqueue time of one MAE training step (how far ahead of the GPU the Python thread runs).
python tools/host_enqueue_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import param_groups
from ssl4gie_amd.Models.mae import models_mae

dev = torch.device("cuda:0")
torch.manual_seed(0)
model = models_mae.mae_vit_base_patch16(norm_pix_loss=True).to(dev).set_precision("bf16")
opt = torch.optim.AdamW(param_groups(model), lr=1.5e-4, betas=(0.9, 0.95), fused=True)
imgs = torch.randn(256, 3, 224, 224).to(dev)


def step():
    opt.zero_grad(set_to_none=True)
    loss, _, _ = model(imgs, mask_ratio=0.75)
    loss.backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
hs, ws = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    hs.append((t1 - t0) * 1e3); ws.append((t2 - t0) * 1e3)
print(f"host enqueue {sorted(hs)[5]:.2f} ms of {sorted(ws)[5]:.2f} ms per step (median of 10)")
