"""DPT dense-prediction decoder on the MI355X engine — drop-in for the reference class
(`/root/reference/Models/DPT_decoder.py:315-539`): same constructor signature, same module tree
and therefore the same state_dict keys (`act_postprocess12.0.weight`, `layer1_rn.weight`,
`refinenet1.resConfUnit1.conv1.weight`, `output_conv.0.weight`, ...; 20.07 M parameters for the
depth variant), same `forward(activations) -> [B, 1, 224, 224]`.

The `nn.Conv2d` / `nn.ConvTranspose2d` children only hold parameters; the arithmetic runs on
libssl4gie_hip.so through `ssl4gie_amd.dpt_engine` with channels-last maps (the ViT's token-major
rows ARE the 14x14 channels-last map, so `Slice -> Transpose -> Unflatten` costs one cast).
Both heads are built with `use_readout="ignore"` (the reference's default, models.py:408):
  * dense="depth": fusion blocks without BatchNorm, head 3x3 -> x2 -> 3x3 -> ReLU -> 1x1 -> Sigmoid;
  * dense="seg" (reference :461,483-497): BatchNorm2d after both convolutions of
    every ResidualConvUnit (convolutions without bias), head 3x3(no bias) -> BatchNorm2d -> ReLU ->
    Dropout(0.1) -> 1x1 -> bilinear x2, output [B, num_classes, 224, 224] logits.  BatchNorm runs on
    the channels-last rows with the ResNet BatchNorm kernels (ReLU and the unit's residual add
    fused; SyncBatchNorm-capable); Dropout is torch's own op inside the autograd graph (it draws
    from torch's RNG exactly like the reference's nn.Dropout).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..dpt_engine import (AddFn, Conv3x3Fn, ForkFn, ConvTransposeFn, DepthHeadFn, SegHeadFn, TokensToMapFn,
                          Upsample2xFn)
from ..engine import EngineModule, LinearFn
from ..resnet_engine import BatchNormFn


class _RCU(nn.Module):
    """ResidualConvUnit_custom container: conv1, conv2 3x3 (bias = not bn) [+ bn1, bn2]."""

    def __init__(self, features, bn=False):
        super().__init__()
        self.bn = bn
        self.conv1 = nn.Conv2d(features, features, 3, 1, 1, bias=not bn)
        self.conv2 = nn.Conv2d(features, features, 3, 1, 1, bias=not bn)
        if bn:
            self.bn1 = nn.BatchNorm2d(features)
            self.bn2 = nn.BatchNorm2d(features)


class _Fusion(nn.Module):
    """FeatureFusionBlock_custom container: out_conv 1x1, resConfUnit1, resConfUnit2."""

    def __init__(self, features, bn=False):
        super().__init__()
        self.out_conv = nn.Conv2d(features, features, 1, 1, 0, bias=True)
        self.resConfUnit1 = _RCU(features, bn)
        self.resConfUnit2 = _RCU(features, bn)


class DPT_decoder(EngineModule):
    def __init__(self, num_classes=1, dense="seg", vit_features=768, features=[96, 192, 384, 768],
                 fusion_features=256, use_readout="ignore", size=[224, 224], patch_size=[16, 16]):
        super().__init__()
        if dense not in ("depth", "seg"):
            raise ValueError(f"dense must be 'depth' or 'seg', got {dense!r}")
        if use_readout != "ignore":
            raise NotImplementedError("only use_readout='ignore' (the reference's default) is built")
        f = features
        # parameter-free readout stages keep their slots so that Sequential indices (= state_dict
        # keys) match: act_postprocessN1 has no tensors
        self.act_postprocess12 = nn.Sequential(nn.Conv2d(vit_features, f[0], 1),
                                               nn.ConvTranspose2d(f[0], f[0], 4, 4, 0, bias=True))
        self.act_postprocess22 = nn.Sequential(nn.Conv2d(vit_features, f[1], 1),
                                               nn.ConvTranspose2d(f[1], f[1], 2, 2, 0, bias=True))
        self.act_postprocess32 = nn.Sequential(nn.Conv2d(vit_features, f[2], 1))
        self.act_postprocess42 = nn.Sequential(nn.Conv2d(vit_features, f[3], 1),
                                               nn.Conv2d(f[3], f[3], 3, 2, 1))
        self.layer1_rn = nn.Conv2d(f[0], fusion_features, 3, 1, 1, bias=False)
        self.layer2_rn = nn.Conv2d(f[1], fusion_features, 3, 1, 1, bias=False)
        self.layer3_rn = nn.Conv2d(f[2], fusion_features, 3, 1, 1, bias=False)
        self.layer4_rn = nn.Conv2d(f[3], fusion_features, 3, 1, 1, bias=False)
        use_bn = dense == "seg"
        self.refinenet1 = _Fusion(fusion_features, use_bn)
        self.refinenet2 = _Fusion(fusion_features, use_bn)
        self.refinenet3 = _Fusion(fusion_features, use_bn)
        self.refinenet4 = _Fusion(fusion_features, use_bn)
        if dense == "depth":
            # indices 1 (Interpolate), 3 (ReLU), 5 (Sigmoid) are parameter-free
            self.output_conv = nn.Sequential(nn.Conv2d(fusion_features, fusion_features // 2, 3, 1, 1),
                                             nn.Identity(),
                                             nn.Conv2d(fusion_features // 2, 32, 3, 1, 1),
                                             nn.Identity(),
                                             nn.Conv2d(32, 1, 1, 1, 0),
                                             nn.Identity())
        else:
            # indices 2 (ReLU), 3 (Dropout), 5 (Interpolate) are parameter-free
            self.output_conv = nn.Sequential(nn.Conv2d(fusion_features, fusion_features, 3, 1, 1, bias=False),
                                             nn.BatchNorm2d(fusion_features),
                                             nn.Identity(),
                                             nn.Dropout(0.1, False),
                                             nn.Conv2d(fusion_features, num_classes, 1),
                                             nn.Identity())
        self.dense = dense
        self.grid = (size[0] // patch_size[0], size[1] // patch_size[1])

    # ------------------------------------------------------------------ building blocks
    def _lin(self, x2, conv: nn.Conv2d):
        """Conv2d(k=1) on channels-last rows = Linear."""
        return LinearFn.apply(x2, conv.weight, conv.bias, self.dtype_, self.dtype_, self.sink(),
                              self.lp_cache)

    def _c3(self, x, conv: nn.Conv2d, relu_in=False):
        return Conv3x3Fn.apply(x, conv.weight, conv.bias, conv.stride[0], relu_in, self.sink(),
                               self.lp_cache)

    def _c3_stats(self, x, conv: nn.Conv2d, relu_in=False):
        """bias-free convolution in front of a BatchNorm: (map, partial batch statistics or None)"""
        return Conv3x3Fn.apply(x, conv.weight, conv.bias, conv.stride[0], relu_in, self.sink(),
                               self.lp_cache, True)

    def _bn(self, x, bn, relu=False, res=None, stats=None):
        """BatchNorm2d over the rows of a channels-last map (+ residual) (+ ReLU)"""
        shp = x.shape
        C = shp[-1]
        r2 = res.contiguous().view(-1, C) if res is not None else None
        y = BatchNormFn.apply(x.contiguous().view(-1, C), bn.weight, bn.bias, r2, bn, relu, self.sink(),
                              stats)
        return y.view(shp)

    def _rcu(self, x, rcu: _RCU):
        """out = [bn2](conv2(relu([bn1](conv1(relu(x)))))) + x  (reference :212-233)"""
        x, skip = ForkFn.apply(x)  # two consumers: the gradients meet in the library's add kernel, not autograd's
        if rcu.bn:
            out, st = self._c3_stats(x, rcu.conv1, relu_in=True)
            out = self._bn(out, rcu.bn1, relu=True, stats=st)
            out, st = self._c3_stats(out, rcu.conv2)
            return self._bn(out, rcu.bn2, res=skip, stats=st)  # the skip add rides on bn2
        out = self._c3(x, rcu.conv1, relu_in=True)
        out = self._c3(out, rcu.conv2, relu_in=True)
        return AddFn.apply(out, skip)

    def _fusion(self, blk: _Fusion, x0, x1=None):
        """reference :281-301; refinenet4 gets one input, so its resConfUnit1 never runs"""
        out = x0
        if x1 is not None:
            out = AddFn.apply(out, self._rcu(x1, blk.resConfUnit1))
        out = self._rcu(out, blk.resConfUnit2)
        out = Upsample2xFn.apply(out)
        B, H, W, C = out.shape
        return self._lin(out.view(-1, C), blk.out_conv).view(B, H, W, -1)

    # ------------------------------------------------------------------ forward
    def forward_skip(self, activations):
        """4 x fp32 [B, 1+L, D] taps -> channels-last maps [B,56,56,256] ... [B,7,7,256]"""
        self._prepare()
        gh, gw = self.grid
        B = activations[0].shape[0]
        t = [TokensToMapFn.apply(z, self.dtype_) for z in activations]
        a12, a22, a32, a42 = (self.act_postprocess12, self.act_postprocess22,
                              self.act_postprocess32, self.act_postprocess42)
        l1 = ConvTransposeFn.apply(self._lin(t[0], a12[0]), a12[1].weight, a12[1].bias, B, gh, gw,
                                   self.sink(), self.lp_cache)
        l2 = ConvTransposeFn.apply(self._lin(t[1], a22[0]), a22[1].weight, a22[1].bias, B, gh, gw,
                                   self.sink(), self.lp_cache)
        l3 = self._lin(t[2], a32[0]).view(B, gh, gw, -1)
        l4 = self._c3(self._lin(t[3], a42[0]).view(B, gh, gw, -1), a42[1])
        return [self._c3(l1, self.layer1_rn), self._c3(l2, self.layer2_rn),
                self._c3(l3, self.layer3_rn), self._c3(l4, self.layer4_rn)]

    def forward(self, activations):
        l1, l2, l3, l4 = self.forward_skip(activations)
        p4 = self._fusion(self.refinenet4, l4)
        p3 = self._fusion(self.refinenet3, p4, l3)
        p2 = self._fusion(self.refinenet2, p3, l2)
        p1 = self._fusion(self.refinenet1, p2, l1)
        oc = self.output_conv
        if self.dense == "seg":
            h, st = self._c3_stats(p1, oc[0])
            h = self._bn(h, oc[1], relu=True, stats=st)
            h = nn.functional.dropout(h, oc[3].p, self.training)
            return SegHeadFn.apply(h, oc[4].weight, oc[4].bias, self.sink(), self.lp_cache)
        h = self._c3(p1, oc[0])
        h = Upsample2xFn.apply(h)
        h = self._c3(h, oc[2])
        return DepthHeadFn.apply(h, oc[4].weight, oc[4].bias, self.sink())
