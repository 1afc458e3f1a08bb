#!/usr/bin/env python3
"""Which Python lines issue device memcpys in a MoCo step?  (torch.profiler with stacks)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functools import partial
import torch
from torch.profiler import profile, ProfilerActivity
from ssl4gie_amd.Models.moco_v3.moco import builder
from ssl4gie_amd.Models.resnet import resnet50
from ssl4gie_amd.optim import ArenaLARS
torch.manual_seed(0)
m = builder.MoCo_ResNet(partial(resnet50, zero_init_residual=True), 256, 4096, 1.0).cuda().set_precision("bf16")
opt = ArenaLARS(m, [p for p in m.parameters() if p.requires_grad], lr=0.03, weight_decay=1e-6, momentum=0.9)
x1 = torch.randn(32, 3, 224, 224, device="cuda"); x2 = torch.randn(32, 3, 224, 224, device="cuda")
def step():
    opt.zero_grad(set_to_none=True)
    loss = m(x1, x2, 0.99); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::_to_copy", "aten::to", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous"):
        st = [s for s in (ev.stack or []) if "ssl4gie_amd" in s or "bench" in s]
        cnt[(ev.name, st[0] if st else "?")] += 1
for (name, where), c in cnt.most_common(25):
    print(f"{c:5d}  {name:16s} {where}")
