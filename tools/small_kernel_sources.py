"""Attribute the small torch kernels (fills, copies, adds) of one MAE training step to Python call
sites with torch.profiler.  python tools/small_kernel_sources.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
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


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ev = prof.key_averages(group_by_stack_n=6)
rows = [e for e in ev if e.key in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::add_", "aten::empty_like",
                                   "aten::zeros_like", "aten::ones_like", "aten::_to_copy")]
rows.sort(key=lambda e: -e.count)
for e in rows[:25]:
    print(f"{e.key:18s} count={e.count:4d} cuda_us={e.device_time_total:8.1f}")
    for s in e.stack[:6]:
        print("      ", s[-110:])
