import sys; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
from oracle import synth, moco_ref, resnet_ref
g = np.load("tests/golden/g14_moco_curve.npz")
keys = g["keys"].tolist()
# shapes from the engine-free route: build via torchvision restatement
from oracle import torchvision_restatement as tvr, timm_restatement
from functools import partial
import importlib.util, os
tvr.install_as_torchvision()
import torchvision.models as tvm
spec = importlib.util.spec_from_file_location("ref_builder", "/root/reference/Models/moco_v3/moco/builder.py")
ref_b = importlib.util.module_from_spec(spec); spec.loader.exec_module(ref_b)
torch.manual_seed(0)
m = ref_b.MoCo_ResNet(partial(tvm.resnet50, zero_init_residual=True), 256, 1024, 1.0)
own = m.state_dict()
sd32 = synth.keyed_state_dict({k: tuple(v.shape) for k, v in own.items()}, 61)
for k in list(sd32):
    if k.startswith("momentum_encoder."):
        kb = "base_encoder." + k[len("momentum_encoder."):]
        if kb in sd32 and "running" not in k and "num_batches" not in k: sd32[k] = sd32[kb].clone()
sd = {k: (v.double().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd32.items()}
x1, x2 = synth.moco_views(b=16, size=128)[0]
x1, x2 = x1.double(), x2.double()
def enc(prefix, x):
    sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    return moco_ref.mlp_forward(sub, "fc.", resnet_ref.resnet50_pooled(sub, x))
pred = {k[len("predictor."):]: v for k, v in sd.items() if k.startswith("predictor.")}
q1 = moco_ref.mlp_forward(pred, "", enc("base_encoder.", x1)); q2 = moco_ref.mlp_forward(pred, "", enc("base_encoder.", x2))
with torch.no_grad():
    k1, k2 = enc("momentum_encoder.", x1), enc("momentum_encoder.", x2)
loss = moco_ref.contrastive_loss(q1, k2, 1.0) + moco_ref.contrastive_loss(q2, k1, 1.0)
loss.backward()
print("fp64 oracle loss", float(loss), "reference fp32 loss", g["losses"][0])
names = g["step0/grad_names"].tolist(); norms = dict(zip(names, g["step0/grad_norms"].tolist()))
rows = []
for k in names:
    t = sd[k].grad
    if f"step0/grad/{k}" in g.files:
        r = torch.from_numpy(g[f"step0/grad/{k}"]).double(); e = float((t - r).norm() / (t.norm() + 1e-30))
    else:
        r = torch.from_numpy(g[f"step0/gslice/{k}"]).double(); got = t.reshape(t.shape[0], -1)[:8, :64]
        e = float((got - r).norm() / (got.norm() + 1e-30))
    rows.append((e, k))
rows.sort(reverse=True)
es = np.array([r[0] for r in rows])
print("reference-fp32 (fixture) vs fp64 oracle: median", np.median(es), "p90", np.quantile(es, .9), "max", rows[0], "min", rows[-1])
