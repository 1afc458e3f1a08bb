"""CPU oracle: functional fp32 restatement of torchvision 0.10 ResNet50 (Bottleneck v1.5) in
training mode.  TEST INFRASTRUCTURE.

torchvision is pinned by the reference (`requirements.txt:10`; call sites `Models/models.py:3,63-75`,
`Models/moco_v3/main_moco.py:30,185-187`) but is neither under /root/reference nor installed in this
image, and the reference holds no tests or fixtures for this path: **parity unpinned** — the
restatement follows torchvision's published `resnet.py` (conv1 7x7/2 -> bn1 -> relu -> maxpool 3x3/2
-> layer1..4 of Bottleneck{conv1 1x1, bn1, relu, conv2 3x3 (stride), bn2, relu, conv3 1x1, bn3,
(+ downsample = conv1x1(stride) + bn), relu} -> avgpool) on a state_dict with torchvision's key
names, and is anchored only by torch op semantics (F.conv2d / F.batch_norm / F.max_pool2d).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

LAYERS = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))


def _bn(sd, p, x, eps=1e-5):
    return F.batch_norm(x, None, None, sd[p + ".weight"], sd[p + ".bias"], True, 0.1, eps)


def bottleneck(sd, p, x, stride, has_down):
    out = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"])))
    out = F.relu(_bn(sd, p + ".bn2", F.conv2d(out, sd[p + ".conv2.weight"], stride=stride, padding=1)))
    out = _bn(sd, p + ".bn3", F.conv2d(out, sd[p + ".conv3.weight"]))
    idn = x
    if has_down:
        idn = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride))
    return F.relu(out + idn)


def resnet50_features(sd, imgs):
    x = F.conv2d(imgs, sd["conv1.weight"], stride=2, padding=3)
    x = F.relu(_bn(sd, "bn1", x))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, (planes, blocks, stride) in enumerate(LAYERS, start=1):
        for j in range(blocks):
            x = bottleneck(sd, f"layer{li}.{j}", x, stride if j == 0 else 1, j == 0)
    return x


def resnet50_pooled(sd, imgs):
    return resnet50_features(sd, imgs).mean((2, 3))


# ------------------------------------------------------------------ dense decoder of the ResNet path
# Reference: Models/models.py:14-60 (ResNet_Dec_Block / ResNet_Dec_Level), :88-107 (decoder_levels,
# output_conv), :128-135 (decode).  PINNED: tests/golden/g12_resnet_dec.npz holds outputs and
# gradients of the reference's own `ResNet_from_Any(dense="depth").decode` (models.py imported in
# the authoring container on top of oracle/timm_restatement.py + oracle/torchvision_restatement.py,
# tests/golden/make_golden.py:g12_resnet_dec), and tests/test_oracle_golden.py checks this
# restatement against it.  The trunk above stays unpinned at the torchvision boundary.
def _bnb(sd, p, x, eps=1e-5):
    return F.batch_norm(x, None, None, sd[p + ".weight"], sd[p + ".bias"], True, 0.1, eps)


def dec_block(sd, p, x, fusion):
    identity = x
    if fusion:
        identity = _bnb(sd, p + ".identity.1", F.conv2d(x, sd[p + ".identity.0.weight"], sd[p + ".identity.0.bias"]))
    q = p + ".process"
    out = F.relu(_bnb(sd, q + ".1", F.conv2d(x, sd[q + ".0.weight"], sd[q + ".0.bias"])))
    out = F.relu(_bnb(sd, q + ".4", F.conv2d(out, sd[q + ".3.weight"], sd[q + ".3.bias"], padding=1)))
    out = _bnb(sd, q + ".7", F.conv2d(out, sd[q + ".6.weight"], sd[q + ".6.bias"]))
    return F.relu(out + identity)


def dec_level(sd, p, x_low, x_high, n_blocks=3):
    r = _bnb(sd, p + ".chan_reduce.1", F.conv2d(x_low, sd[p + ".chan_reduce.0.weight"], sd[p + ".chan_reduce.0.bias"]))
    x = torch.cat((F.interpolate(r, scale_factor=2, mode="bilinear", align_corners=True), x_high), 1)
    for j in range(n_blocks):
        x = dec_block(sd, f"{p}.blocks.{j}", x, j == 0)
    return x


def resnet50_stage_maps(sd, imgs):
    x = F.conv2d(imgs, sd["conv1.weight"], stride=2, padding=3)
    x = F.relu(_bn(sd, "bn1", x))
    x = F.max_pool2d(x, 3, 2, 1)
    maps = []
    for li, (planes, blocks, stride) in enumerate(LAYERS, start=1):
        for j in range(blocks):
            x = bottleneck(sd, f"layer{li}.{j}", x, stride if j == 0 else 1, j == 0)
        maps.append(x)
    return maps


def output_head(sd, out):
    """models.py:96-104: up x2 -> 3x3 (256->128) -> up x2 -> 3x3 (128->32) -> ReLU -> 1x1 -> Sigmoid"""
    up = lambda t: F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=True)
    h = F.conv2d(up(out), sd["output_conv.1.weight"], sd["output_conv.1.bias"], padding=1)
    h = F.conv2d(up(h), sd["output_conv.3.weight"], sd["output_conv.3.bias"], padding=1)
    return torch.sigmoid(F.conv2d(F.relu(h), sd["output_conv.5.weight"], sd["output_conv.5.bias"]))


def decode(sd, m):
    """models.py:128-135 on the four stage maps m = [256@s, 512@s/2, 1024@s/4, 2048@s/8]"""
    out = dec_level(sd, "decoder_levels.0", m[-1], m[-2])
    out = dec_level(sd, "decoder_levels.1", out, m[-3])
    out = dec_level(sd, "decoder_levels.2", out, m[-4])
    return output_head(sd, out)


def resnet50_dense(sd, imgs):
    """ResNet_from_Any(dense=...)._forward_impl: stage maps -> 3 decoder levels -> output_conv"""
    return decode(sd, resnet50_stage_maps(sd, imgs))
