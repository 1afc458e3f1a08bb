"""Checkpoint formats either side of the hot path.

The engine's modules use the reference's parameter names, so a state_dict is exchanged unchanged;
what differs between producers is the WRAPPING, which these helpers undo exactly like the
reference's own scripts do:

  * MAE pretraining (`Models/mae/util/misc.py:301-307`): {"model": state_dict, "optimizer": ...}
    -> `ViT_from_MAE(weight_path=...)` reads ["model"] and copies the names it owns
    (`Models/models.py:417-425`);
  * MoCo-v3 pretraining (`Models/moco_v3/main_moco.py:310-316`): {"state_dict": {"module.
    base_encoder.*", "module.momentum_encoder.*", "module.predictor.*"}, ...} -> the converter
    `Models/moco_v3/convert_to_deit.py:24-32` keeps `module.base_encoder.*` without its `head.*`
    (ViT) and strips the prefix; for ResNet50 the projector lives under `fc.*` and is dropped the
    same way (the finetune model has `fc = Identity`, models.py:77-80);
  * Barlow Twins (upstream checkpoints): {"model": {"module.backbone.*", "module.projector.*",
    "module.bn.*"}} or a bare ResNet50 state_dict -> backbone tensors without `fc.*`;
  * finetune checkpoints of this repo (`train_depth.py:355-366`): {"model_state_dict": ...} saved
    from `model.module.state_dict()`.
"""
from __future__ import annotations

from collections import OrderedDict


def load_file(path, map_location="cpu"):
    """`torch.load` for checkpoint files written by the reference's scripts.  Those hold more than
    tensors — MAE saves `'args': argparse.Namespace` (`Models/mae/util/misc.py:301-307`), the finetune
    apps save Python / NumPy RNG states (`train_depth.py:355-366`) — which torch >= 2.6's default
    `weights_only=True` rejects.  First try the restricted unpickler with `argparse.Namespace`
    allow-listed; a file that still needs more (RNG state tuples, numpy arrays) is loaded the way the
    reference's torch did, unrestricted: these are the user's own training checkpoints."""
    import argparse
    import pickle
    import torch
    try:
        with torch.serialization.safe_globals([argparse.Namespace]):
            return torch.load(path, map_location=map_location, weights_only=True)
    except (pickle.UnpicklingError, RuntimeError):
        return torch.load(path, map_location=map_location, weights_only=False)


def _unwrap(obj):
    """the tensor dict inside any of the wrappers above"""
    if isinstance(obj, dict):
        for key in ("state_dict", "model", "model_state_dict"):
            if key in obj and isinstance(obj[key], dict):
                return obj[key]
    return obj


def strip_prefix(state_dict, prefix):
    return OrderedDict((k[len(prefix):], v) for k, v in state_dict.items() if k.startswith(prefix))


def moco_to_backbone(checkpoint, drop=("head.", "fc.")):
    """convert_to_deit.py:24-32: keep `module.base_encoder.*` minus the projector, strip the prefix.
    Accepts the DDP-saved file content or an un-prefixed `MoCo.state_dict()`."""
    sd = _unwrap(checkpoint)
    pre = "module.base_encoder." if any(k.startswith("module.") for k in sd) else "base_encoder."
    out = strip_prefix(sd, pre)
    return OrderedDict((k, v) for k, v in out.items() if not k.startswith(tuple(drop)))


def barlow_twins_to_backbone(checkpoint):
    """ResNet50 / ViT trunk tensors of a Barlow Twins checkpoint (`backbone.*`), or the bare
    torchvision state_dict the upstream project also publishes; `fc.*` dropped."""
    sd = _unwrap(checkpoint)
    for pre in ("module.backbone.", "backbone."):
        if any(k.startswith(pre) for k in sd):
            sd = strip_prefix(sd, pre)
            break
    return OrderedDict((k, v) for k, v in sd.items() if not k.startswith(("fc.", "head.")))


def ddp_unwrap(state_dict):
    """`module.`-prefixed names of a DistributedDataParallel-saved dict -> plain names"""
    sd = _unwrap(state_dict)
    if sd and all(k.startswith("module.") for k in sd):
        return strip_prefix(sd, "module.")
    return sd


def load_matching(model, state_dict, verbose=False):
    """Copy every tensor whose name and shape the model owns (the reference's
    `load_my_state_dict`, models.py:417-425); returns (loaded, missing, unexpected) name lists."""
    import torch
    own = model.state_dict()
    loaded, unexpected = [], []
    with torch.no_grad():
        for name, value in state_dict.items():
            if name in own and tuple(own[name].shape) == tuple(value.shape):
                own[name].copy_(value)
                loaded.append(name)
            else:
                unexpected.append(name)
    missing = [k for k in own if k not in set(loaded)]
    if verbose:
        print(f"Successfully loaded params for {len(loaded)} items")
    return loaded, missing, unexpected
