"""CPU oracle: functional fp32 restatement of the DPT depth decoder.  TEST INFRASTRUCTURE.

Follows `/root/reference/Models/DPT_decoder.py` (forward_skip :501-527, forward :529-539,
FeatureFusionBlock_custom.forward :281-301, ResidualConvUnit_custom.forward :212-233, depth head
:468-482) on a plain state_dict with the reference's key names, NCHW like the reference.  Pinned by
`tests/golden/g6_dpt_depth.npz`, which `tests/golden/make_golden.py` generates by running the
reference's own `DPT_decoder` class (importable as is: pure torch) on the same seeded weights and
inputs (see tests/test_oracle_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def dpt_param_shapes(vit=768, feats=(96, 192, 384, 768), fus=256):
    s = {}
    f = feats
    s["act_postprocess12.0.weight"] = (f[0], vit, 1, 1); s["act_postprocess12.0.bias"] = (f[0],)
    s["act_postprocess12.1.weight"] = (f[0], f[0], 4, 4); s["act_postprocess12.1.bias"] = (f[0],)
    s["act_postprocess22.0.weight"] = (f[1], vit, 1, 1); s["act_postprocess22.0.bias"] = (f[1],)
    s["act_postprocess22.1.weight"] = (f[1], f[1], 2, 2); s["act_postprocess22.1.bias"] = (f[1],)
    s["act_postprocess32.0.weight"] = (f[2], vit, 1, 1); s["act_postprocess32.0.bias"] = (f[2],)
    s["act_postprocess42.0.weight"] = (f[3], vit, 1, 1); s["act_postprocess42.0.bias"] = (f[3],)
    s["act_postprocess42.1.weight"] = (f[3], f[3], 3, 3); s["act_postprocess42.1.bias"] = (f[3],)
    for i in range(4):
        s[f"layer{i + 1}_rn.weight"] = (fus, f[i], 3, 3)
    for i in range(1, 5):
        s[f"refinenet{i}.out_conv.weight"] = (fus, fus, 1, 1); s[f"refinenet{i}.out_conv.bias"] = (fus,)
        for u in (1, 2):
            for c in (1, 2):
                s[f"refinenet{i}.resConfUnit{u}.conv{c}.weight"] = (fus, fus, 3, 3)
                s[f"refinenet{i}.resConfUnit{u}.conv{c}.bias"] = (fus,)
    s["output_conv.0.weight"] = (fus // 2, fus, 3, 3); s["output_conv.0.bias"] = (fus // 2,)
    s["output_conv.2.weight"] = (32, fus // 2, 3, 3); s["output_conv.2.bias"] = (32,)
    s["output_conv.4.weight"] = (1, 32, 1, 1); s["output_conv.4.bias"] = (1,)
    return s


def dpt_state_dict(seed: int, **kw):
    """seeded weights: N(0, 1/sqrt(fan_in)) so activations stay O(1) through the 20-conv chain"""
    g = torch.Generator("cpu").manual_seed(seed)
    sd = {}
    for name, shp in dpt_param_shapes(**kw).items():
        if name.endswith(".bias"):
            sd[name] = 0.1 * torch.randn(shp, generator=g)
        else:
            fan_in = shp[1] * shp[2] * shp[3] if ".1.weight" not in name or "act_postprocess42" in name \
                else shp[0]  # ConvTranspose2d weights are [Cin, Cout, k, k]
            sd[name] = torch.randn(shp, generator=g) / fan_in ** 0.5
    return sd


def _rcu(sd, p, x):
    out = F.conv2d(F.relu(x), sd[p + ".conv1.weight"], sd[p + ".conv1.bias"], padding=1)
    out = F.conv2d(F.relu(out), sd[p + ".conv2.weight"], sd[p + ".conv2.bias"], padding=1)
    return out + x


def _fusion(sd, p, x0, x1=None):
    out = x0
    if x1 is not None:
        out = out + _rcu(sd, p + ".resConfUnit1", x1)
    out = _rcu(sd, p + ".resConfUnit2", out)
    out = F.interpolate(out, scale_factor=2, mode="bilinear", align_corners=True)
    return F.conv2d(out, sd[p + ".out_conv.weight"], sd[p + ".out_conv.bias"])


def dpt_forward(sd, activations, grid=(14, 14), return_all=False):
    """activations: 4 x [B, 1 + L, D] -> [B, 1, 16 gh, 16 gw]"""
    maps = []
    for z in activations:
        B, L1, D = z.shape
        maps.append(z[:, 1:].transpose(1, 2).reshape(B, D, grid[0], grid[1]))
    l1 = F.conv2d(maps[0], sd["act_postprocess12.0.weight"], sd["act_postprocess12.0.bias"])
    l1 = F.conv_transpose2d(l1, sd["act_postprocess12.1.weight"], sd["act_postprocess12.1.bias"], stride=4)
    l2 = F.conv2d(maps[1], sd["act_postprocess22.0.weight"], sd["act_postprocess22.0.bias"])
    l2 = F.conv_transpose2d(l2, sd["act_postprocess22.1.weight"], sd["act_postprocess22.1.bias"], stride=2)
    l3 = F.conv2d(maps[2], sd["act_postprocess32.0.weight"], sd["act_postprocess32.0.bias"])
    l4 = F.conv2d(maps[3], sd["act_postprocess42.0.weight"], sd["act_postprocess42.0.bias"])
    l4 = F.conv2d(l4, sd["act_postprocess42.1.weight"], sd["act_postprocess42.1.bias"], stride=2, padding=1)
    l1 = F.conv2d(l1, sd["layer1_rn.weight"], None, padding=1)
    l2 = F.conv2d(l2, sd["layer2_rn.weight"], None, padding=1)
    l3 = F.conv2d(l3, sd["layer3_rn.weight"], None, padding=1)
    l4 = F.conv2d(l4, sd["layer4_rn.weight"], None, padding=1)
    p4 = _fusion(sd, "refinenet4", l4)
    p3 = _fusion(sd, "refinenet3", p4, l3)
    p2 = _fusion(sd, "refinenet2", p3, l2)
    p1 = _fusion(sd, "refinenet1", p2, l1)
    h = F.conv2d(p1, sd["output_conv.0.weight"], sd["output_conv.0.bias"], padding=1)
    h = F.interpolate(h, scale_factor=2, mode="bilinear", align_corners=True)
    h = F.conv2d(h, sd["output_conv.2.weight"], sd["output_conv.2.bias"], padding=1)
    out = torch.sigmoid(F.conv2d(F.relu(h), sd["output_conv.4.weight"], sd["output_conv.4.bias"]))
    if return_all:
        return out, {"layer_4": l4, "path_4": p4, "path_1": p1}
    return out


def ssi_loss(prediction, target, alpha=0.1, scales=4):
    """Depth_estimation/Metrics/losses.py:120-146 (+ :5-25, :28-38, :51-77, :104-117)"""
    pred = prediction.squeeze(1)
    tgt = target.squeeze(1)
    mask = (tgt > 0).to(pred.dtype)
    a00 = (mask * pred * pred).sum((1, 2)); a01 = (mask * pred).sum((1, 2)); a11 = mask.sum((1, 2))
    b0 = (mask * pred * tgt).sum((1, 2)); b1 = (mask * tgt).sum((1, 2))
    det = a00 * a11 - a01 * a01
    x0 = torch.zeros_like(b0); x1 = torch.zeros_like(b1)
    v = det != 0
    x0[v] = (a11[v] * b0[v] - a01[v] * b1[v]) / det[v]
    x1[v] = (-a01[v] * b0[v] + a00[v] * b1[v]) / det[v]
    ssi = x0.view(-1, 1, 1) * pred + x1.view(-1, 1, 1)

    def batch(il, M):
        d = M.sum()
        return il.sum() / d if d != 0 else il.sum() * 0

    total = batch((mask * (ssi - tgt) ** 2).sum((1, 2)), 2 * mask.sum((1, 2)))
    reg = 0
    for s in range(scales):
        st = 2 ** s
        p, t, m = ssi[:, ::st, ::st], tgt[:, ::st, ::st], mask[:, ::st, ::st]
        diff = m * (p - t)
        gx = (diff[:, :, 1:] - diff[:, :, :-1]).abs() * (m[:, :, 1:] * m[:, :, :-1])
        gy = (diff[:, 1:, :] - diff[:, :-1, :]).abs() * (m[:, 1:, :] * m[:, :-1, :])
        reg = reg + batch(gx.sum((1, 2)) + gy.sum((1, 2)), m.sum((1, 2)))
    return total + alpha * reg


# ------------------------------------------------------------------ "seg" variant
# Reference: DPT_decoder.py:461 (use_bn = dense == "seg"), ResidualConvUnit_custom :166-233 (convs
# without bias, BatchNorm2d after each), output_conv :483-497.  Pinned by tests/golden/g9_dpt_seg.npz
# (generated from the reference's own class with Dropout.p set to 0: its mask is torch-RNG state,
# not arithmetic).
def seg_param_shapes(num_classes=1, vit=768, feats=(96, 192, 384, 768), fus=256):
    s = {k: v for k, v in dpt_param_shapes(vit, feats, fus).items()
         if not k.startswith("output_conv") and ".resConfUnit" not in k}
    for i in range(1, 5):
        for u in (1, 2):
            for c in (1, 2):
                s[f"refinenet{i}.resConfUnit{u}.conv{c}.weight"] = (fus, fus, 3, 3)
                s[f"refinenet{i}.resConfUnit{u}.bn{c}.weight"] = (fus,)
                s[f"refinenet{i}.resConfUnit{u}.bn{c}.bias"] = (fus,)
    s["output_conv.0.weight"] = (fus, fus, 3, 3)
    s["output_conv.1.weight"] = (fus,); s["output_conv.1.bias"] = (fus,)
    s["output_conv.4.weight"] = (num_classes, fus, 1, 1); s["output_conv.4.bias"] = (num_classes,)
    return s


def seg_state_dict(seed: int, **kw):
    g = torch.Generator("cpu").manual_seed(seed)
    sd = {}
    for name, shp in seg_param_shapes(**kw).items():
        if len(shp) == 1:
            is_gamma = name.endswith(".weight")
            sd[name] = (1.0 + 0.1 * torch.randn(shp, generator=g)) if is_gamma else 0.1 * torch.randn(shp, generator=g)
        else:
            fan_in = shp[1] * shp[2] * shp[3] if ".1.weight" not in name or "act_postprocess42" in name \
                else shp[0]
            sd[name] = torch.randn(shp, generator=g) / fan_in ** 0.5
    return sd


def _bn2d(sd, p, x, eps=1e-5):
    return F.batch_norm(x, None, None, sd[p + ".weight"], sd[p + ".bias"], True, 0.1, eps)


def _rcu_bn(sd, p, x):
    out = _bn2d(sd, p + ".bn1", F.conv2d(F.relu(x), sd[p + ".conv1.weight"], None, padding=1))
    out = _bn2d(sd, p + ".bn2", F.conv2d(F.relu(out), sd[p + ".conv2.weight"], None, padding=1))
    return out + x


def _fusion_bn(sd, p, x0, x1=None):
    out = x0
    if x1 is not None:
        out = out + _rcu_bn(sd, p + ".resConfUnit1", x1)
    out = _rcu_bn(sd, p + ".resConfUnit2", out)
    out = F.interpolate(out, scale_factor=2, mode="bilinear", align_corners=True)
    return F.conv2d(out, sd[p + ".out_conv.weight"], sd[p + ".out_conv.bias"])


def seg_forward(sd, activations, grid=(14, 14)):
    """activations: 4 x [B, 1 + L, D] -> logits [B, num_classes, 16 gh, 16 gw] (training-mode
    BatchNorm, Dropout inactive)"""
    maps = []
    for z in activations:
        B, L1, D = z.shape
        maps.append(z[:, 1:].transpose(1, 2).reshape(B, D, grid[0], grid[1]))
    l1 = F.conv2d(maps[0], sd["act_postprocess12.0.weight"], sd["act_postprocess12.0.bias"])
    l1 = F.conv_transpose2d(l1, sd["act_postprocess12.1.weight"], sd["act_postprocess12.1.bias"], stride=4)
    l2 = F.conv2d(maps[1], sd["act_postprocess22.0.weight"], sd["act_postprocess22.0.bias"])
    l2 = F.conv_transpose2d(l2, sd["act_postprocess22.1.weight"], sd["act_postprocess22.1.bias"], stride=2)
    l3 = F.conv2d(maps[2], sd["act_postprocess32.0.weight"], sd["act_postprocess32.0.bias"])
    l4 = F.conv2d(maps[3], sd["act_postprocess42.0.weight"], sd["act_postprocess42.0.bias"])
    l4 = F.conv2d(l4, sd["act_postprocess42.1.weight"], sd["act_postprocess42.1.bias"], stride=2, padding=1)
    l1 = F.conv2d(l1, sd["layer1_rn.weight"], None, padding=1)
    l2 = F.conv2d(l2, sd["layer2_rn.weight"], None, padding=1)
    l3 = F.conv2d(l3, sd["layer3_rn.weight"], None, padding=1)
    l4 = F.conv2d(l4, sd["layer4_rn.weight"], None, padding=1)
    p4 = _fusion_bn(sd, "refinenet4", l4)
    p3 = _fusion_bn(sd, "refinenet3", p4, l3)
    p2 = _fusion_bn(sd, "refinenet2", p3, l2)
    p1 = _fusion_bn(sd, "refinenet1", p2, l1)
    h = F.relu(_bn2d(sd, "output_conv.1", F.conv2d(p1, sd["output_conv.0.weight"], None, padding=1)))
    h = F.conv2d(h, sd["output_conv.4.weight"], sd["output_conv.4.bias"])
    return F.interpolate(h, scale_factor=2, mode="bilinear", align_corners=True)


def soft_dice_loss(logits, targets, smooth=1e-8):
    """Binary_segmentation/Metrics/losses.py:5-24"""
    num = targets.size(0)
    m1 = torch.sigmoid(logits).view(num, -1)
    m2 = targets.view(num, -1)
    score = 2.0 * ((m1 * m2).sum(1) + smooth) / ((m1 * m1).sum(1) + (m2 * m2).sum(1) + smooth)
    return 1 - score.sum() / num
