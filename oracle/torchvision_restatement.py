"""CPU restatement of the torchvision==0.10.0 classes the reference subclasses.  TEST INFRASTRUCTURE.

The reference (`/root/reference/requirements.txt:10`) pins torchvision 0.10.0 without vendoring it
and torchvision is not installed in this image.  `Models/models.py:3,63-69` derives
`ResNet_from_Any` from `torchvision.models.resnet.ResNet(Bottleneck, [3, 4, 6, 3])` and overrides
`_forward_impl` (:137-152); `Models/moco_v3/main_moco.py:185-187` builds
`torchvision_models.resnet50(zero_init_residual=True)`.  These classes restate torchvision's
*published* `resnet.py` for exactly that use — attribute names, module order and hence state_dict
keys identical (`conv1, bn1, relu, maxpool, layer1..4, avgpool, fc`; `Bottleneck.{conv1,bn1,conv2,
bn2,conv3,bn3,relu,downsample}`, stride on the 3x3 = v1.5; kaiming-normal fan_out init; BN weight 1 /
bias 0; `zero_init_residual` -> bn3.weight = 0) — so that the reference's own `Models/models.py` can
be imported on top of them when golden vectors are generated (`tests/golden/make_golden.py`,
fixtures G12).  The decoder arithmetic in those fixtures (`ResNet_Dec_Block/Level`, `decode`,
models.py:16-60,128-135) is the reference's own code; the trunk arithmetic is this restatement
(parity at the torchvision boundary is pinned by torch op semantics only, like `resnet_ref.py`).

Only tests/golden/make_golden.py installs this (in the authoring container).
"""
from __future__ import annotations

import torch
import torch.nn as nn


def conv3x3(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=False)


def conv1x1(cin, cout, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=1, stride=stride, bias=False)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64,
                 dilation=1, norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        assert groups == 1 and base_width == 64 and dilation == 1, "not used by the reference"
        width = planes
        self.conv1 = conv1x1(inplanes, width)
        self.bn1 = norm_layer(width)
        self.conv2 = conv3x3(width, width, stride)
        self.bn2 = norm_layer(width)
        self.conv3 = conv1x1(width, planes * self.expansion)
        self.bn3 = norm_layer(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        identity = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            identity = self.downsample(x)
        out = out + identity
        return self.relu(out)


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000, zero_init_residual=False, groups=1,
                 width_per_group=64, replace_stride_with_dilation=None, norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        self._norm_layer = norm_layer
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, self.inplanes, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = norm_layer(self.inplanes)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(conv1x1(self.inplanes, planes * block.expansion, stride),
                                       self._norm_layer(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample, norm_layer=self._norm_layer)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, norm_layer=self._norm_layer))
        return nn.Sequential(*layers)

    def _forward_impl(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))

    def forward(self, x):
        return self._forward_impl(x)


def resnet50(**kw):
    return ResNet(Bottleneck, [3, 4, 6, 3], **kw)


def install_as_torchvision():
    """Register this module under the `torchvision.*` names `Models/models.py` touches."""
    import sys
    import types

    me = sys.modules[__name__]
    pkgs = {}
    for name in ("torchvision", "torchvision.models", "torchvision.models.resnet",
                 "torchvision.models.utils", "torchvision.ops"):
        m = types.ModuleType(name)
        m.__path__ = []
        pkgs[name] = m
        sys.modules[name] = m
    pkgs["torchvision"].__version__ = "0.10.0"
    for n in ("ResNet", "Bottleneck", "resnet50"):
        setattr(pkgs["torchvision.models.resnet"], n, getattr(me, n))
    pkgs["torchvision.models"].resnet50 = resnet50

    def _no_network(*a, **k):
        raise RuntimeError("no network in this environment")

    pkgs["torchvision.models.utils"].load_state_dict_from_url = _no_network
    pkgs["torchvision"].models = pkgs["torchvision.models"]
    pkgs["torchvision"].ops = pkgs["torchvision.ops"]
    pkgs["torchvision.models"].resnet = pkgs["torchvision.models.resnet"]
    pkgs["torchvision.models"].utils = pkgs["torchvision.models.utils"]
