"""Are the bf16 operand copies refreshed after an optimizer step?  (torch's fused AdamW does not bump
Tensor._version.)  python tools/check_lp_refresh.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import param_groups
from ssl4gie_amd.Models.mae import models_mae

dev = torch.device("cuda:0")
for fused in (True, False):
    torch.manual_seed(0)
    model = models_mae.mae_vit_base_patch16(norm_pix_loss=True).to(dev).set_precision("bf16")
    opt = torch.optim.AdamW(param_groups(model), lr=1.5e-3, betas=(0.9, 0.95), fused=fused)
    imgs = torch.randn(32, 3, 224, 224, generator=torch.Generator().manual_seed(1)).to(dev)
    w = model.blocks[0].attn.qkv.weight
    v0 = w._version
    losses = []
    for it in range(6):
        opt.zero_grad(set_to_none=True)
        torch.manual_seed(5)
        loss, _, _ = model(imgs, mask_ratio=0.75)
        loss.backward()
        opt.step()
        losses.append(round(float(loss), 5))
    lp, _ = model.lp_cache.get(w, torch.bfloat16) if hasattr(model, "lp_cache") else (None, None)
    stale = None if lp is None else float((lp.float() - w.detach().to(torch.bfloat16).float()).abs().max())
    print(f"fused={fused}: version {v0} -> {w._version}; losses {losses}; max |cached bf16 - bf16(w)| = {stale}")
