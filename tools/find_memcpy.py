#!/usr/bin/env python3
"""Which Python lines issue device copies / fills / torch arithmetic kernels in one training step?
(torch.profiler with stacks).  usage: find_memcpy.py [mae|depth|moco|det]"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functools import partial
import torch
from torch.profiler import profile, ProfilerActivity
which = sys.argv[1] if len(sys.argv) > 1 else "mae"
torch.manual_seed(0)
if which == "moco":
    from ssl4gie_amd.Models.moco_v3.moco import builder
    from ssl4gie_amd.Models.resnet import resnet50
    from ssl4gie_amd.optim import ArenaLARS
    m = builder.MoCo_ResNet(partial(resnet50, zero_init_residual=True), 256, 4096, 1.0).cuda().set_precision("bf16")
    opt = ArenaLARS(m, [p for p in m.parameters() if p.requires_grad], lr=0.03, weight_decay=1e-6, momentum=0.9)
    x1 = torch.randn(32, 3, 224, 224, device="cuda"); x2 = torch.randn(32, 3, 224, 224, device="cuda")
    def step():
        opt.zero_grad(set_to_none=True)
        loss = m(x1, x2, 0.99); loss.backward(); opt.step()
elif which == "det":
    from ssl4gie_amd.Models import models
    from ssl4gie_amd.optim import ArenaAdamW
    m = models.VisionTransformer_from_Any(False, 0, False, None, True, 1024, 768, 12, 12, "cls").cuda().set_precision("bf16")
    opt = ArenaAdamW(m, [p for p in m.parameters() if p.requires_grad], lr=1e-4)
    x = torch.randn(2, 3, 1024, 1024, device="cuda")
    def step():
        opt.zero_grad(set_to_none=True)
        out = m(x)
        loss = sum((v * v).mean() for v in out.values()); loss.backward(); opt.step()
elif which == "depth":
    from ssl4gie_amd.Models import models
    from ssl4gie_amd.losses import ScaleAndShiftInvariantLoss
    from ssl4gie_amd.optim import ArenaAdamW
    m = models.ViT_from_MAE(None, False, 1, False, "depth", False, None, 768, 12, 12, "cls").cuda().set_precision("bf16")
    opt = ArenaAdamW(m, list(m.parameters()), lr=1e-4)
    x = torch.randn(16, 3, 224, 224, device="cuda"); t = torch.rand(16, 1, 224, 224, device="cuda")
    lf = ScaleAndShiftInvariantLoss(alpha=0.1)
    def step():
        opt.zero_grad(set_to_none=True)
        loss = lf(m(x), t); loss.backward(); opt.step()
else:
    from ssl4gie_amd.Models.mae import models_mae
    from ssl4gie_amd.optim import ArenaAdamW
    m = models_mae.mae_vit_base_patch16(norm_pix_loss=True).cuda().set_precision("bf16")
    opt = ArenaAdamW(m, list(m.parameters()), lr=1.5e-4, betas=(0.9, 0.95))
    x = torch.randn(64, 3, 224, 224, device="cuda")
    def step():
        opt.zero_grad(set_to_none=True)
        loss, _, _ = m(x, mask_ratio=0.75); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
# Python-level attribution: wrap the Tensor methods that copy and record the first ssl4gie_amd frame
import traceback
calls = collections.Counter()
def wrap(name):
    orig = getattr(torch.Tensor, name)
    def f(self, *a, **k):
        if name != "contiguous" or not self.is_contiguous():
            if name != "to" or (a and (isinstance(a[0], (torch.dtype, torch.device, str)) or torch.is_tensor(a[0]))) or k:
                fr = [x for x in traceback.extract_stack()[:-1] if "ssl4gie_amd" in x.filename]
                where = f"{os.path.relpath(fr[-1].filename)}:{fr[-1].lineno} {fr[-1].line}" if fr else "?"
                calls[(name, tuple(self.shape), str(self.dtype).replace("torch.", ""), where)] += 1
        return orig(self, *a, **k)
    setattr(torch.Tensor, name, f)
    return orig
origs = {n: wrap(n) for n in ("copy_", "clone", "contiguous", "to", "float", "bfloat16", "add_", "zero_", "fill_")}
step(); torch.cuda.synchronize()
for n, o in origs.items(): setattr(torch.Tensor, n, o)
print(f"== {which}: Python-level Tensor copies / in-place arithmetic in ONE step")
for (name, shape, dt, where), c in sorted(calls.items(), key=lambda x: -x[1])[:50]:
    print(f"{c:4d} {name:11s} {str(shape):24s} {dt:9s} {where[:150]}")
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.name not in ("aten::empty", "aten::empty_like", "aten::empty_strided", "aten::view",
            "aten::reshape", "aten::as_strided", "aten::select", "aten::slice", "aten::detach", "aten::alias", "aten::t",
            "aten::transpose", "aten::permute", "aten::unsqueeze", "aten::squeeze", "aten::expand", "aten::narrow",
            "aten::_unsafe_view", "aten::result_type", "aten::is_nonzero", "aten::item", "aten::_local_scalar_dense",
            "aten::resize_", "aten::set_", "aten::flatten", "aten::unflatten", "aten::view_as", "aten::lift_fresh", "aten::detach_"):
        st = [s for s in (ev.stack or []) if "ssl4gie_amd" in s or "bench" in s or "find_memcpy" in s]
        cnt[(ev.name, st[0] if st else "?")] += 1
print(f"== {which}: aten ops that may launch kernels / copies in ONE step, by first ssl4gie_amd frame")
for (name, where), c in cnt.most_common(60):
    print(f"{c:5d}  {name:22s} {where}")
# device-side view: kernels and memcpys of the step
dev = collections.Counter()
for ev in prof.events():
    if ev.device_type is not None and str(ev.device_type).endswith("CUDA"):
        dev[ev.name[:80]] += 1
print("== device activities")
for n, c in dev.most_common(40):
    if "Memcpy" in n or "copy" in n.lower() or "at::native" in n or "fill" in n.lower() or "Memset" in n:
        print(f"{c:5d}  {n}")
