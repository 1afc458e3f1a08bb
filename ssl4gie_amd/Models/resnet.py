"""ResNet50 trunk on the MI355X engine with torchvision's module tree / state_dict names
(`conv1.weight, bn1.{weight,bias,running_mean,running_var,num_batches_tracked},
layer{1-4}.{j}.{conv1,bn1,conv2,bn2,conv3,bn3}.*, layer{k}.0.downsample.{0,1}.*`).

torchvision 0.10 is pinned by the reference (`requirements.txt:10`) but neither vendored nor
installed, so this restates `ResNet(Bottleneck, [3, 4, 6, 3])` (v1.5: the stride sits on the 3x3
convolution; kaiming-normal fan_out init; BN weight 1 / bias 0; `zero_init_residual` zeroes every
bn3.weight — used by MoCo, Models/moco_v3/main_moco.py:186).  The nn.Conv2d / nn.BatchNorm2d
children only hold parameters and buffers; the arithmetic runs on libssl4gie_hip.so
(`ssl4gie_amd.resnet_engine`), channels-last.
"""
from __future__ import annotations

import torch
import torch.nn as nn

import os

from ..dpt_engine import Conv3x3Fn

# SSL4GIE_GRAD_JOIN=0: autograd sums the two gradient contributions of a block input itself
_GRAD_JOIN = os.environ.get("SSL4GIE_GRAD_JOIN", "1") != "0"
# SSL4GIE_BN_STATS_FUSED=0: BatchNorm statistics by their own pass over the map (A/B measurements)
_BN_STATS = os.environ.get("SSL4GIE_BN_STATS_FUSED", "1") != "0"
# SSL4GIE_BN_RECOMPUTE=0: the 1x1 convolutions that widen a map 4x (conv3, downsample) write their raw output and
# a BatchNorm pass re-reads it under torch.no_grad() too (A/B measurements)
_BN_RECOMPUTE = os.environ.get("SSL4GIE_BN_RECOMPUTE", "1") != "0"
from ..engine import EngineModule, GradJoin, LinearFn
from ..resnet_engine import (AvgPoolFn, BatchNormFn, BnReluConv3x3Fn, BnReluMaxPoolFn, MaxPoolFn, StemConvFn,
                             Subsample2Fn, _count_batch, _sync_group, bn_relu_conv3x3_ok, bn_relu_maxpool_ok)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride


class ResNet50(EngineModule):
    """trunk only: `forward_features` -> channels-last layer4 map (or the 4 stage maps)"""

    def __init__(self, zero_init_residual=False, num_classes=None):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(64, 3)
        self.layer2 = self._make_layer(128, 4, stride=2)
        self.layer3 = self._make_layer(256, 6, stride=2)
        self.layer4 = self._make_layer(512, 3, stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        if num_classes is not None:  # torchvision's classifier slot (MoCo replaces it by its projector)
            self.fc = nn.Linear(2048, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)

    def _make_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride, bias=False),
                                       nn.BatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(Bottleneck(self.inplanes, planes))
        return nn.Sequential(*layers)

    # ------------------------------------------------------------------ engine forward
    def _bn(self, x, bn, relu, res=None, stats=None, join=None):
        return BatchNormFn.apply(x, bn.weight, bn.bias, res, bn, relu, self.sink(), stats, join)

    def _c1(self, x, conv, join=None):
        """1x1 convolution -> (map, BatchNorm partial statistics of the map or None)"""
        if conv.stride[0] == 2:
            x = Subsample2Fn.apply(x, join)  # the pixel pick deposits its scattered gradient in the join
            join = None  # ... and sits between x and the GEMM: the GEMM's gradient is not x's
        B, H, W, C = x.shape
        r = LinearFn.apply(x.reshape(-1, C), conv.weight, None, self.dtype_, self.dtype_, self.sink(),
                           self.lp_cache, _BN_STATS, join)
        y, st = r if _BN_STATS else (r, None)
        return y.view(B, H, W, -1), st

    def _c1_bn_nograd(self, x, conv, bn, relu, res=None):
        """1x1 convolution + training-mode BatchNorm (+ residual) (+ ReLU) when nothing is kept for a backward pass
        (torch.no_grad(): MoCo's momentum encoder, moco/builder.py:127-135) and the map gets WIDER (conv3,
        downsample: C -> 4C): the product runs twice — first for the batch statistics alone (colstats with C == NULL:
        nothing written), then with the normalisation, the residual add and the ReLU in its epilogue — instead of
        writing the raw 4C-wide map and re-reading it in a BatchNorm pass: 2.5 instead of 4.25 passes over that map.
        The statistics are those of the bf16-rounded raw outputs, as on the other path; the affine map is applied to
        the fp32 accumulators (one rounding less).  Returns None when the shapes do not qualify."""
        from .. import ops
        if conv.stride[0] == 2:
            x = ops.subsample2(x.contiguous())
        B, H, W, C = x.shape
        n_out = conv.weight.shape[0]
        x2 = x.reshape(-1, C)
        w, _ = self.lp_cache.get(conv.weight, self.dtype_)
        stats = ops.linear_colstats_only(x2, w)
        mom = bn.momentum if bn.momentum is not None else 0.1
        gd = bn.weight.detach() if bn.weight is not None else None
        bd = bn.bias.detach() if bn.bias is not None else None
        sync, group = _sync_group(bn)
        if sync:   # SyncBatchNorm: the statistics-only product's partials -> exchange -> global coefficients
            from ..resnet_engine import sync_batch_stats
            mean_l, var_l = ops.bn_stats_from_partials(stats, x2.shape[0])   # of the product's n_out columns
            mean, rstd, _ = sync_batch_stats(mean_l, var_l, x2.shape[0], bn, group)
            coef = ops.bn_coef_stats(mean, rstd, gd, bd)
        else:
            coef, _, _ = ops.bn_coef_partials(stats, x2.shape[0], gd, bd, bn.running_mean, bn.running_var, mom, bn.eps)
        if bn.num_batches_tracked is not None:
            _count_batch(bn)
        r2 = res.contiguous().view(-1, n_out) if res is not None else None
        return ops.linear_affine_fwd(x2, w, coef[0], coef[1], r2, relu).view(B, H, W, n_out)

    def _recompute_ok(self, x, conv, bn):
        from .. import ops
        if not _BN_RECOMPUTE or torch.is_grad_enabled() or self.dtype_ != torch.bfloat16:
            return False
        from ..resnet_engine import _SYNC_FUSED
        if not (bn.training or bn.running_mean is None) or (_sync_group(bn)[0] and not _SYNC_FUSED):
            return False
        st = conv.stride[0]
        rows = x.shape[0] * ((x.shape[1] - 1) // st + 1) * ((x.shape[2] - 1) // st + 1)
        return conv.weight.shape[0] > conv.weight.shape[1] and \
            ops.colstats_ok(rows, conv.weight.shape[0], conv.weight.shape[1], self.dtype_)

    def _block(self, x, blk: Bottleneck):
        # every convolution hands the batch statistics of its output to the BatchNorm that follows
        # (column sums from the GEMM epilogue): no separate statistics pass over the maps
        # x feeds conv1 and the identity (or downsample) branch: one GradJoin, conv1 is the adder
        join = GradJoin.for_tensor(x) if _GRAD_JOIN else None
        out, st = self._c1(x, blk.conv1, join)
        if bn_relu_conv3x3_ok(out, blk.bn1, st, blk.conv2, torch.is_grad_enabled()):
            # bn1 -> relu applied inside the 3x3 convolution's halo staging: no BatchNorm pass, no normalised map
            out, st = BnReluConv3x3Fn.apply(out, blk.bn1.weight, blk.bn1.bias, blk.bn1, st, blk.conv2.weight,
                                            self.sink(), self.lp_cache, _BN_STATS)
        else:
            out = self._bn(out, blk.bn1, True, stats=st)
            r = Conv3x3Fn.apply(out, blk.conv2.weight, None, blk.conv2.stride[0], False, self.sink(),
                                self.lp_cache, _BN_STATS)
            out, st = r if _BN_STATS else (r, None)
        out = self._bn(out, blk.bn2, True, stats=st)
        identity = x
        if blk.downsample is not None:
            if self._recompute_ok(x, blk.downsample[0], blk.downsample[1]):
                identity = self._c1_bn_nograd(x, blk.downsample[0], blk.downsample[1], False)
            else:
                idn, sti = self._c1(x, blk.downsample[0], join)
                identity = self._bn(idn, blk.downsample[1], False, stats=sti)
            join = None  # bn3's residual input is the downsample branch, not x
        if self._recompute_ok(out, blk.conv3, blk.bn3):
            return self._c1_bn_nograd(out, blk.conv3, blk.bn3, True, res=identity)
        out, st = self._c1(out, blk.conv3)
        return self._bn(out, blk.bn3, True, res=identity, stats=st, join=join)  # relu(bn3(out) + identity)

    def prepare_inputs(self, imgs):
        """build the per-batch stem operand (bf16: the packed image) on the CURRENT stream, so that two
        forward passes over the same batch on different streams only read it (moco/builder.py)"""
        from .. import ops
        from ..resnet_engine import STEM_COLS, _STEM_DIRECT
        if imgs.dtype != torch.float32 or not imgs.is_contiguous():
            return  # StemConvFn makes its own copy: nothing is shared between the passes
        if _STEM_DIRECT and self.dtype_ == torch.bfloat16 and self.conv1.weight.shape[0] == 64 and imgs.shape[1] == 3:
            STEM_COLS.get(imgs, self.dtype_, ops.stem7x7_pack)
        else:
            STEM_COLS.get(imgs, self.dtype_)   # the patch matrix (fp32 parity mode)

    def forward_maps(self, imgs, all_stages=False):
        self._prepare()
        r = StemConvFn.apply(imgs, self.conv1.weight, self.dtype_, self.sink(), self.lp_cache, _BN_STATS)
        x, st = r if _BN_STATS else (r, None)
        if bn_relu_maxpool_ok(x, self.bn1, st):   # bn1 -> relu -> maxpool in one pass over the convolution output
            x = BnReluMaxPoolFn.apply(x, self.bn1.weight, self.bn1.bias, self.bn1, self.sink(), st)
        else:
            x = self._bn(x, self.bn1, True, stats=st)
            x = MaxPoolFn.apply(x)
        maps = []
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                x = self._block(x, blk)
            maps.append(x)
        return maps if all_stages else x

    def pooled(self, imgs):
        """[B, 2048] fp32: avgpool + flatten of the layer4 map"""
        return AvgPoolFn.apply(self.forward_maps(imgs))


def resnet50(num_classes=1000, zero_init_residual=False, **kwargs):
    """torchvision.models.resnet50 signature subset used by the reference (main_moco.py:185-187)"""
    return ResNet50(zero_init_residual=zero_init_residual, num_classes=num_classes)
