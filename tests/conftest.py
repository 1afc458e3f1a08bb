import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(a, b):
    """max |a-b| relative to the scale of the reference tensor b."""
    import torch
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    den = b.abs().max().item()
    return (a - b).abs().max().item() / (den if den > 0 else 1.0)


def keyed_weights(module, seed, keys=None, digest=None, keep=()):
    """Load oracle.synth.keyed_tensor(key, shape, seed) into one of the build's modules (the same
    tensors tests/golden/make_golden.py loaded into the reference's class; `keep` = fixed tables
    the module builds itself) and prove it: same key set, same SHA-256 over the loaded state_dict.
    Returns the state_dict as plain tensors (for the oracle)."""
    import torch
    from oracle import synth
    own = module.state_dict()
    if keys is not None:
        assert sorted(own) == sorted(np.asarray(keys).tolist()), \
            set(own) ^ set(np.asarray(keys).tolist())
    sd = synth.keyed_state_dict({k: tuple(v.shape) for k, v in own.items()}, seed, keep=keep)
    missing, unexpected = module.load_state_dict(sd, strict=False)   # (a state_dict entry may be a copy: ViTDet_FPN)
    assert set(missing) <= set(keep) and not unexpected, (missing, unexpected)
    full = {k: v.detach().clone().cpu() for k, v in module.state_dict().items()}
    if digest is not None:
        assert synth.state_dict_digest(full) == str(digest), "weights differ from the generator's"
    return full
