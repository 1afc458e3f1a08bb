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
    from `model.module.state_dict()`;
  * ImageNet "augreg" ViT-B/16 weights (`Models/models.py:286-290`): a Flax `.npz` that the reference
    downloads and hands to timm's `VisionTransformer.load_pretrained` -> `_load_weights`.  timm 0.6.12
    is pinned (`requirements.txt:7`) but not vendored, so `augreg_npz_to_state_dict` restates its
    published mapping (names, HWIO -> OIHW, [in, heads, hd] -> [out, in], q/k/v concatenation);
    there is no network here: the file is read from a local path.
"""
from __future__ import annotations

from collections import OrderedDict


def _reference_payload_globals():
    """what the reference's own checkpoint files hold besides tensors and builtins: MAE's
    `'args': argparse.Namespace` (`Models/mae/util/misc.py:301-307`), the finetune apps' NumPy RNG
    state and NumPy scalars (`train_depth.py:355-366`: `np.random.get_state()`, `loss` / `val_perf`)
    — under the module paths NumPy 1.x and 2.x pickle them with"""
    import argparse
    import numpy as np
    try:
        from numpy._core import multiarray as ma
    except ImportError:  # NumPy 1.x
        from numpy.core import multiarray as ma
    out = [argparse.Namespace, np.ndarray, np.dtype]
    for name in ("_reconstruct", "scalar"):
        fn = getattr(ma, name)
        out += [fn, (fn, f"numpy.core.multiarray.{name}"), (fn, f"numpy._core.multiarray.{name}")]
    for t in (np.uint8, np.uint32, np.uint64, np.int32, np.int64, np.float16, np.float32, np.float64, np.bool_):
        out.append(type(np.dtype(t)))
    return out


def load_file(path, map_location="cpu", trusted=None):
    """`torch.load` for checkpoint files written by the reference's scripts.  Those hold more than
    tensors — MAE saves `'args': argparse.Namespace`, the finetune apps save Python / NumPy RNG states —
    which torch >= 2.6's default `weights_only=True` rejects.  The restricted unpickler is kept and
    exactly those types are allow-listed (`_reference_payload_globals`); nothing else is ever executed.
    A file that needs more is REFUSED unless the caller opts in with `trusted=True` (or
    `SSL4GIE_TRUSTED_CHECKPOINTS=1`): a downloaded "pretrained weights" file must not be able to run
    code just because the safe loader rejected it."""
    import os
    import pickle
    import warnings
    import torch
    try:
        with torch.serialization.safe_globals(_reference_payload_globals()):
            return torch.load(path, map_location=map_location, weights_only=True)
    except (pickle.UnpicklingError, RuntimeError, AttributeError, TypeError) as e:
        # AttributeError / TypeError: torch < 2.6 has no `safe_globals` or rejects the (callable, "module.path")
        # entries of the allow-list — the same refusal / opt-in path applies there (torch >= 2.6 is the supported
        # floor for the restricted load)
        if trusted is None:
            trusted = os.environ.get("SSL4GIE_TRUSTED_CHECKPOINTS", "0") == "1"
        if not trusted:
            raise RuntimeError(
                f"{path}: holds objects outside tensors and the reference's known checkpoint payload "
                f"(argparse.Namespace, NumPy arrays / scalars): {e}\nIf this file is your own, load it with "
                "checkpoints.load_file(path, trusted=True) or set SSL4GIE_TRUSTED_CHECKPOINTS=1 "
                "(full unpickling can execute code).") from e
        warnings.warn(f"{path}: unpickling without restrictions (trusted=True)")
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


# ------------------------------------------------------------------ timm "augreg" .npz (Flax) weights
def _n2p(w, t=True):
    """timm 0.6.12 vision_transformer._load_weights._n2p: numpy (Flax layout) -> torch layout"""
    import numpy as np
    import torch
    w = np.asarray(w)
    if w.ndim == 4 and w.shape[0] == w.shape[1] == w.shape[2] == 1:
        w = w.flatten()
    if t:
        if w.ndim == 4:
            w = w.transpose([3, 2, 0, 1])   # HWIO -> OIHW
        elif w.ndim == 3:
            w = w.transpose([2, 0, 1])
        elif w.ndim == 2:
            w = w.transpose([1, 0])
    return torch.from_numpy(np.ascontiguousarray(w))


def augreg_npz_to_state_dict(npz, depth=12, prefix=""):
    """Flax ViT checkpoint (dict-like of numpy arrays, e.g. numpy.load(path)) -> state_dict with the
    timm / reference names.  Mapping of timm 0.6.12 `_load_weights` for the plain (non-hybrid) ViT:
    `embedding/{kernel,bias}` -> patch_embed.proj, `cls` -> cls_token,
    `Transformer/posembed_input/pos_embedding` -> pos_embed, `Transformer/encoder_norm/{scale,bias}` ->
    norm, `Transformer/encoderblock_i/{LayerNorm_0, MultiHeadDotProductAttention_1/{query,key,value,out},
    LayerNorm_2, MlpBlock_3/Dense_{0,1}}` -> blocks.i.{norm1, attn.qkv / attn.proj, norm2, mlp.fc1 / fc2}.
    The classifier (`head/*`, `pre_logits/*`) is not mapped: the reference replaces it (models.py:291)."""
    import torch
    w = npz
    sd = OrderedDict()
    sd["patch_embed.proj.weight"] = _n2p(w[f"{prefix}embedding/kernel"])
    sd["patch_embed.proj.bias"] = _n2p(w[f"{prefix}embedding/bias"])
    sd["cls_token"] = _n2p(w[f"{prefix}cls"], t=False)
    sd["pos_embed"] = _n2p(w[f"{prefix}Transformer/posembed_input/pos_embedding"], t=False)
    sd["norm.weight"] = _n2p(w[f"{prefix}Transformer/encoder_norm/scale"])
    sd["norm.bias"] = _n2p(w[f"{prefix}Transformer/encoder_norm/bias"])
    for i in range(depth):
        bp = f"{prefix}Transformer/encoderblock_{i}/"
        mha = bp + "MultiHeadDotProductAttention_1/"
        b = f"blocks.{i}."
        sd[b + "norm1.weight"] = _n2p(w[bp + "LayerNorm_0/scale"])
        sd[b + "norm1.bias"] = _n2p(w[bp + "LayerNorm_0/bias"])
        sd[b + "attn.qkv.weight"] = torch.cat(
            [_n2p(w[mha + n + "/kernel"], t=False).flatten(1).T for n in ("query", "key", "value")])
        sd[b + "attn.qkv.bias"] = torch.cat(
            [_n2p(w[mha + n + "/bias"], t=False).reshape(-1) for n in ("query", "key", "value")])
        sd[b + "attn.proj.weight"] = _n2p(w[mha + "out/kernel"]).flatten(1)
        sd[b + "attn.proj.bias"] = _n2p(w[mha + "out/bias"])
        sd[b + "norm2.weight"] = _n2p(w[bp + "LayerNorm_2/scale"])
        sd[b + "norm2.bias"] = _n2p(w[bp + "LayerNorm_2/bias"])
        for r in range(2):
            sd[b + f"mlp.fc{r + 1}.weight"] = _n2p(w[bp + f"MlpBlock_3/Dense_{r}/kernel"])
            sd[b + f"mlp.fc{r + 1}.bias"] = _n2p(w[bp + f"MlpBlock_3/Dense_{r}/bias"])
    return OrderedDict((k, v.contiguous().float()) for k, v in sd.items())


def state_dict_to_augreg_npz(state_dict, heads=12):
    """inverse of augreg_npz_to_state_dict (dict of numpy arrays in the Flax layout): lets a model of
    this repo be exported to the file format the reference downloads, and pins the loader in the tests"""
    import numpy as np
    sd = {k: v.detach().cpu().float().numpy() for k, v in state_dict.items()}
    D = sd["norm.weight"].shape[0]
    hd = D // heads
    out = {
        "embedding/kernel": sd["patch_embed.proj.weight"].transpose(2, 3, 1, 0),  # OIHW -> HWIO
        "embedding/bias": sd["patch_embed.proj.bias"],
        "cls": sd["cls_token"],
        "Transformer/posembed_input/pos_embedding": sd["pos_embed"],
        "Transformer/encoder_norm/scale": sd["norm.weight"],
        "Transformer/encoder_norm/bias": sd["norm.bias"],
    }
    i = 0
    while f"blocks.{i}.norm1.weight" in sd:
        b, bp = f"blocks.{i}.", f"Transformer/encoderblock_{i}/"
        mha = bp + "MultiHeadDotProductAttention_1/"
        out[bp + "LayerNorm_0/scale"], out[bp + "LayerNorm_0/bias"] = sd[b + "norm1.weight"], sd[b + "norm1.bias"]
        out[bp + "LayerNorm_2/scale"], out[bp + "LayerNorm_2/bias"] = sd[b + "norm2.weight"], sd[b + "norm2.bias"]
        for j, n in enumerate(("query", "key", "value")):
            wj = sd[b + "attn.qkv.weight"][j * D:(j + 1) * D]            # [out, in]
            out[mha + n + "/kernel"] = wj.T.reshape(D, heads, hd)          # [in, heads, hd]
            out[mha + n + "/bias"] = sd[b + "attn.qkv.bias"][j * D:(j + 1) * D].reshape(heads, hd)
        out[mha + "out/kernel"] = sd[b + "attn.proj.weight"].reshape(D, heads, hd).transpose(1, 2, 0)
        out[mha + "out/bias"] = sd[b + "attn.proj.bias"]
        for r in range(2):
            out[bp + f"MlpBlock_3/Dense_{r}/kernel"] = sd[b + f"mlp.fc{r + 1}.weight"].T
            out[bp + f"MlpBlock_3/Dense_{r}/bias"] = sd[b + f"mlp.fc{r + 1}.bias"]
        i += 1
    return {k: np.ascontiguousarray(v) for k, v in out.items()}


def load_augreg_npz(model, path):
    """`VisionTransformer.load_pretrained(npz)` of the reference's ImageNet path (models.py:286-290):
    copies the trunk tensors of a Flax ViT-B/16 checkpoint into `model` (the pos_embed must already
    have the model's token count: 197 for the 224 x 224 / patch 16 file the reference names)."""
    import numpy as np
    with np.load(path) as npz:
        depth = 0
        while f"Transformer/encoderblock_{depth}/LayerNorm_0/scale" in npz:
            depth += 1
        sd = augreg_npz_to_state_dict(npz, depth)
    own = model.state_dict()
    for k, v in sd.items():
        if k in own and tuple(own[k].shape) != tuple(v.shape):
            raise ValueError(f"{k}: checkpoint {tuple(v.shape)} vs model {tuple(own[k].shape)}")
    return load_matching(model, sd)


def contiguous_state_dict(module):
    """state_dict() with every entry contiguous.  Since round 6 ViTDet_FPN's own state_dict hook already returns
    contiguous copies of its channels-last (C, H, W) LayerNorm tables (Models/models.py), so this is `state_dict()`
    for every module of the package; kept for callers of the round-5 API and for foreign modules."""
    return {k: (v if v.is_contiguous() else v.contiguous()) for k, v in module.state_dict().items()}
