"""CPU: the C-ABI library loads and exports every symbol include/ssl4gie_hip.h declares (no compute
calls), and host-side logic (state_dict schema, pos tables, arena, gradient sink, LDS swizzles)."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import ROOT, load_golden

from ssl4gie_amd import _lib, engine


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ssl4gie_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ssl4gie_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    names = _declared_symbols()
    assert len(names) >= 25
    assert os.path.exists(_lib.LIB_PATH), "build with __graft_entry__.build()"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ssl4gie_hip.h but not exported"
    assert set(names) == set(_lib.PROTOTYPES), set(names) ^ set(_lib.PROTOTYPES)
    assert _lib.load().ssl4gie_abi_version() == _lib.ABI_VERSION


def test_graft_entry_build_runs_clean():
    """the documented build command (README) must exit 0: make is a no-op when up to date"""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.build()


def test_workspace_queries_need_no_gpu():
    L = _lib.load()
    assert L.ssl4gie_layernorm_bwd_workspace_bytes(50432, 512) == 1024 * 2 * 512 * 4
    d = _lib.BlockDims(256, 197, 512, 16, 2048, _lib.BF16, 1e-6)
    assert L.ssl4gie_block_workspace_bytes(ctypes.byref(d)) > 0
    assert L.ssl4gie_attn_workspace_bytes(_lib.BF16, 8, 197, 12, 64) == 0
    assert L.ssl4gie_attn_workspace_bytes(_lib.F32, 2, 50, 12, 64) == 2 * 2 * 12 * 50 * 50 * 4


def test_ops_refuse_cpu_tensors():
    from ssl4gie_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.layernorm_fwd(torch.zeros(4, 8), torch.ones(8), torch.zeros(8), 1e-6, torch.float32)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libssl4gie_hip.so")
    with pytest.raises(_lib.HipExtensionMissing):
        _lib.load()


def test_mae_state_dict_schema_matches_reference():
    from oracle import mae_ref, synth
    from ssl4gie_amd.Models.mae import models_mae
    m = models_mae.mae_vit_base_patch16(norm_pix_loss=True)
    sd = m.state_dict()
    ref = synth.mae_shapes(mae_ref.VIT_B)
    assert set(sd) == set(ref)
    assert all(tuple(sd[k].shape) == tuple(ref[k]) for k in ref)
    g = load_golden("g5_mae_vitb.npz")
    assert len(sd) == int(g["n_tensors"]) and sum(v.numel() for v in sd.values()) == int(g["n_params"])
    assert sorted(n for n, p in m.named_parameters() if p.requires_grad) == sorted(g["grad_names"])


def test_pos_tables_match_reference_fixture():
    from ssl4gie_amd.Models.mae.util.pos_embed import get_2d_sincos_pos_embed
    g = load_golden("g3_sincos.npz")
    for d in (768, 512, 192, 128):
        np.testing.assert_allclose(get_2d_sincos_pos_embed(d, 14, True).astype(np.float32),
                                   g[f"mae_{d}"], rtol=0, atol=1e-6)
    from oracle import mae_ref
    from ssl4gie_amd.Models.models import moco_sincos_pos_embed
    assert torch.equal(moco_sincos_pos_embed(768, (14, 14)), mae_ref.sincos_2d_moco(768, 14))


def test_finetune_factories_keep_reference_signatures():
    from ssl4gie_amd import utils
    v = utils.get_MAE_backbone(None, True, 6, False, None, False)
    keys = set(v.state_dict())
    assert {"cls_token", "pos_embed", "decoder_pos_embed", "patch_embed.proj.weight",
            "blocks.11.mlp.fc2.bias", "norm.weight", "lin_head.weight"} <= keys
    assert not any(k.startswith("decoder_blocks") or k.startswith("mask_token") for k in keys)
    assert len(keys) == 153
    v2 = utils.get_ImageNet_or_random_ViT(False, None, False, None, False, False)
    assert len(v2.state_dict()) == 150
    v3 = utils.get_MoCoV3_backbone(None, "vit_b", True, 12, False, None, False)
    assert v3.lin_head.weight.shape == (12, 768) and not v3.pos_embed.requires_grad


def test_param_arena_and_grad_sink_cpu():
    lin = nn.Sequential(nn.Linear(8, 16), nn.LayerNorm(16), nn.Linear(16, 4))
    ps = list(lin.parameters())
    before = [p.detach().clone() for p in ps]
    a = engine.ParamArena(ps)
    assert a.intact() and all(torch.equal(p, b) for p, b in zip(ps, before))
    assert all(p.data_ptr() % 256 == a.data.data_ptr() % 256 for p in ps)
    lo, hi = a.span(list(lin[2].parameters()))
    assert hi == a.numel and lo == a.span([ps[4]])[0]
    sink = engine.GradSink(a)
    tg, acc, rets = sink.plan(ps)
    assert not acc and all(t.data_ptr() == a.grad_view(p).data_ptr() for t, p in zip(tg, ps))
    for p, t in zip(ps, rets):
        p.grad = t
    tg2, acc2, rets2 = sink.plan(ps)
    assert acc2 and all(r is None for r in rets2)
    ps[0].grad = None
    with pytest.raises(RuntimeError, match="mixed"):
        sink.plan(ps)
    ps[1].requires_grad = False
    for p in ps:
        p.grad = None
    tg3, _, rets3 = sink.plan(ps)
    assert tg3[1] is None and rets3[1] is None and tg3[0] is not None


def test_channels_last_layernorm_tables_keep_their_layout_cpu():
    """ViTDet_FPN stores its (C, H, W) LayerNorm tables channels-last (Models/models.py); state_dict entries keep
    the reference's shapes and values, and ParamArena re-homes parameter AND gradient with the same strides, so the
    kernels' [H, W, C] views alias the arena memory (no permuted copies: profiles/r04dm)"""
    import copy
    from ssl4gie_amd.Models.models import ViTDet_FPN
    torch.manual_seed(0)
    f = ViTDet_FPN(grid=8, dim=64, out=32)
    tables = [p for m in f.modules() if isinstance(m, nn.LayerNorm) for p in (m.weight, m.bias)]
    assert len(tables) == 18 and all(p.ndim == 3 for p in tables)
    assert all(not p.is_contiguous() and p.permute(1, 2, 0).is_contiguous() for p in tables)
    sd = {k: torch.randn(v.shape) for k, v in f.state_dict().items()}  # contiguous, as a reference checkpoint holds them
    f.load_state_dict(sd)
    assert all(torch.equal(v, sd[k]) and v.shape == sd[k].shape for k, v in f.state_dict().items())
    g = copy.deepcopy(f)  # the layout is recognised by strides: it survives deepcopy / .to()
    for mod in (f, g):
        ps = list(mod.parameters())
        a = engine.ParamArena(ps)
        assert a.intact() and all(torch.equal(v, sd[k]) for k, v in mod.state_dict().items())
        for p in ps:
            gv = a.grad_view(p)
            assert gv.shape == p.shape and gv.stride() == p.stride()
            if p.ndim == 3:
                hwc = p.detach().permute(1, 2, 0)
                assert hwc.is_contiguous() and hwc.data_ptr() == p.data_ptr()
                assert gv.permute(1, 2, 0).is_contiguous()
        # a write through the [H, W, C] view of the gradient slice is the gradient in the parameter's own shape
        w = [p for p in ps if p.ndim == 3][0]
        gv = a.grad_view(w)
        gv.permute(1, 2, 0).copy_(torch.arange(w.numel(), dtype=torch.float32).view(w.shape[1], w.shape[2], w.shape[0]))
        assert float(gv[3, 1, 2]) == float((1 * w.shape[2] + 2) * w.shape[0] + 3)
    # state_dict() itself is all-contiguous (the module's hook copies the 18 tables): safetensors takes
    # it as it is, `.view(-1)` works, and a round trip through the bytes restores values AND the parameter layout
    from safetensors.torch import load as st_load, save as st_save
    sd2 = f.state_dict()
    assert all(v.is_contiguous() for v in sd2.values()) and sd2["fpn1.2.weight"].view(-1).numel() == 32 * 4 * 4
    back = st_load(st_save(sd2))
    h = ViTDet_FPN(grid=8, dim=64, out=32)
    h.load_state_dict(back)
    assert all(torch.equal(v, sd[k]) for k, v in h.state_dict().items())
    assert all(not p.is_contiguous() and p.permute(1, 2, 0).is_contiguous()
               for m in h.modules() if isinstance(m, nn.LayerNorm) for p in (m.weight, m.bias))


def test_lds_swizzles_are_conflict_free():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lds_bank_sim as sim
    nt = lambda r: (r >> 1) & 7
    assert all(sim.b128_cycles(sim.row_read_addrs(128, nt, c0, r0)) == 4
               for c0 in (0, 4) for r0 in (0, 16, 64, 112))
    tn = lambda k: ((k & 3) | ((k >> 1) & 4)) << 1
    assert all(sim.tr_b16_cycles(sim.tr_read_addrs(256, tn, k0, x0, s)) == 2
               for k0 in (0, 32) for x0 in range(0, 128, 16) for s in (False, True))
    for rb, swz, nch in ((128, lambda r: r & 6, 8), (64, lambda r: (r >> 1) & 2, 4)):
        assert all(sim.b128_cycles(sim.row_read_addrs(rb, swz, c0, r0)) == 4
                   for c0 in range(0, nch, 4) for r0 in (0, 16, 208))
        assert all(sim.tr_b16_cycles(sim.attn_tr_read_addrs(rb, swz, k0, d0, s)) == 2
                   for k0 in (0, 32, 192) for d0 in range(0, nch * 8, 16) for s in (False, True))
    # direct 3x3 convolution: every tap read of the three tile geometries, the weight reads and the
    # staged output tile take their ideal cycle counts (confirmed on the GPU: SQ_LDS_BANK_CONFLICT = 0)
    for geom in ((1, 8, 32), (1, 16, 16), (4, 8, 8)):
        assert sim.direct_x_read_cycles(*geom) == 4, geom
    assert sim.direct_w_read_cycles(2) == 4
    assert sim.direct_out_cycles(1) == (2, 4) and sim.direct_out_cycles(2) == (2, 4)


def test_checkpoint_wrappers_round_trip():
    """MoCo / Barlow Twins / DDP wrappers -> backbone state_dicts (host logic only)"""
    from functools import partial
    import torch
    from ssl4gie_amd import checkpoints as ck
    from ssl4gie_amd.Models import models
    from ssl4gie_amd.Models.barlow_twins import BarlowTwins
    from ssl4gie_amd.Models.moco_v3 import vits
    from ssl4gie_amd.Models.moco_v3.moco import builder
    from ssl4gie_amd.Models.resnet import resnet50
    torch.manual_seed(0)
    moco = builder.MoCo_ResNet(partial(resnet50, zero_init_residual=True), 256, 4096, 1.0)
    saved = {"epoch": 3, "state_dict": {"module." + k: v for k, v in moco.state_dict().items()}}
    bb = ck.moco_to_backbone(saved)
    assert not any(k.startswith(("fc.", "head.")) for k in bb) and "conv1.weight" in bb
    target = models.ResNet_from_Any(None, False, 0, False, None)
    assert set(bb) == set(target.state_dict())            # strict-loadable, as the reference expects
    target.load_state_dict(bb, strict=True)
    assert torch.equal(target.layer3[2].conv2.weight, moco.base_encoder.layer3[2].conv2.weight)

    vit = vits.VisionTransformerMoCo(embed_dim=192, depth=2, num_heads=3, num_classes=8)
    del vit.head
    bt = BarlowTwins(vit, 192, "64-64-64")
    bsd = ck.barlow_twins_to_backbone({"model": {"module." + k: v for k, v in bt.state_dict().items()}})
    assert set(bsd) == set(vit.state_dict())

    ddp = {"model_state_dict": {"module." + k: v for k, v in target.state_dict().items()}}
    assert set(ck.ddp_unwrap(ddp)) == set(target.state_dict())
    loaded, missing, unexpected = ck.load_matching(target, {**bb, "extra.weight": torch.zeros(1)})
    assert len(loaded) == len(bb) and missing == [] and unexpected == ["extra.weight"]


def test_window_permutation_matches_reference_construction_cpu():
    """perm / inv_perm of WindowedAttention (models.py:179-191) for N = 4096, window 16:
    integer work, exact — the engine's index construction against the oracle's statement-for-statement
    restatement, plus the structural properties the detection trunk relies on"""
    import torch
    from oracle import det_ref
    from ssl4gie_amd.Models.models import WINDOWED_BLOCKS, window_permutation
    for s in (64, 48, 32, 16):
        perm, inv, windows = det_ref.window_perm(s * s, 16)
        p2, i2 = window_permutation(s, 16)
        assert torch.equal(perm, p2) and torch.equal(inv, i2) and windows == (s // 16) ** 2
        assert torch.equal(perm[inv], torch.arange(s * s)) and torch.equal(inv[perm], torch.arange(s * s))
        # window w = (wi, wj) occupies the contiguous run [256 w, 256 (w + 1)) in row-major order
        for w in (0, windows - 1):
            wi, wj = divmod(w, s // 16)
            blk = perm[256 * w:256 * (w + 1)].reshape(16, 16)
            want = (torch.arange(16)[:, None] + 16 * wi) * s + torch.arange(16)[None, :] + 16 * wj
            assert torch.equal(blk, want)
    assert WINDOWED_BLOCKS == det_ref.WINDOWED == (0, 1, 3, 4, 6, 7, 9, 10)


def test_checkpoint_files_with_non_tensor_payload_load(tmp_path):
    """the reference's MAE checkpoints carry `'args': argparse.Namespace` (mae/util/misc.py:301-307)
    and its finetune checkpoints RNG states (train_depth.py:355-366): torch >= 2.6's default
    weights_only unpickler rejects both; checkpoints.load_file reads them"""
    import argparse
    import random
    from ssl4gie_amd import checkpoints
    sd = {"cls_token": torch.randn(1, 1, 8), "blocks.0.norm1.weight": torch.ones(8)}
    f1 = tmp_path / "mae.pth"
    torch.save({"model": sd, "optimizer": {}, "epoch": 3, "scaler": {}, "args": argparse.Namespace(lr=1e-3)}, f1)
    with pytest.raises(Exception):
        torch.load(f1, map_location="cpu")  # what the round-1 code did
    ck = checkpoints.load_file(str(f1))
    assert ck["epoch"] == 3 and torch.equal(ck["model"]["cls_token"], sd["cls_token"]) and ck["args"].lr == 1e-3
    f2 = tmp_path / "finetune.pth"
    torch.save({"model_state_dict": sd, "py_state": random.getstate(), "np_state": np.random.get_state(),
                "torch_state": torch.get_rng_state()}, f2)
    torch.save({"model_state_dict": sd, "py_state": random.getstate(), "np_state": np.random.get_state(),
                "torch_state": torch.get_rng_state(), "loss": np.float64(0.25), "val_perf": np.float32(1.5)}, f2)
    ck = checkpoints.load_file(str(f2))
    assert torch.equal(checkpoints._unwrap(ck)["blocks.0.norm1.weight"], sd["blocks.0.norm1.weight"])
    assert float(ck["loss"]) == 0.25 and np.array_equal(ck["np_state"][1], np.random.get_state()[1])
    # anything beyond the reference's known payload is refused, not silently unpickled:
    # only an explicit opt-in runs the unrestricted loader
    import fractions
    f3 = tmp_path / "other.pth"
    torch.save({"model": sd, "extra": {"a": fractions.Fraction(1, 1)}}, f3)
    with pytest.raises(RuntimeError, match="trusted"):
        checkpoints.load_file(str(f3))
    with pytest.warns(UserWarning):
        assert checkpoints.load_file(str(f3), trusted=True)["extra"]["a"] == 1
    # an `args` Namespace that carries a pathlib.Path (output_dir=Path(...)) or a list is a realistic MAE payload
    # outside the allow-list: refused with the same message, loadable with the explicit opt-in
    import pathlib
    f4 = tmp_path / "mae_path.pth"
    torch.save({"model": sd, "args": argparse.Namespace(output_dir=pathlib.Path("/tmp/out"), blr=[1e-3, 2e-3])}, f4)
    try:
        ck = checkpoints.load_file(str(f4))           # lists are plain pickle; Path needs the opt-in
        assert ck["args"].blr == [1e-3, 2e-3]
    except RuntimeError as e:
        assert "trusted" in str(e)
        with pytest.warns(UserWarning):
            ck = checkpoints.load_file(str(f4), trusted=True)
        assert ck["args"].output_dir == pathlib.Path("/tmp/out")


def test_augreg_npz_loader_round_trip(tmp_path, monkeypatch):
    """the Flax `.npz` layout the reference's ImageNet path loads through timm
    (models.py:286-290).  Export a seeded trunk to that layout, load it back through the loader the
    `ImageNet_weights=True` constructor uses: every tensor identical; layouts spot-checked against
    timm 0.6.12's published rules (HWIO conv kernel, [in, heads, hd] attention kernels)."""
    from oracle import synth
    from ssl4gie_amd import checkpoints
    from ssl4gie_amd.Models import models
    m = models.VisionTransformer_from_Any(False, 0, False, None, False, None, 768, 12, 12, "cls")
    sd = synth.keyed_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()}, 61)
    flax = checkpoints.state_dict_to_augreg_npz(sd)
    assert flax["embedding/kernel"].shape == (16, 16, 3, 768)
    assert flax["Transformer/encoderblock_3/MultiHeadDotProductAttention_1/key/kernel"].shape == (768, 12, 64)
    assert flax["Transformer/encoderblock_3/MultiHeadDotProductAttention_1/out/kernel"].shape == (12, 64, 768)
    assert flax["Transformer/encoderblock_0/MlpBlock_3/Dense_0/kernel"].shape == (768, 3072)
    # key head h, feature f of the Flax kernel is row D + h*64 + f of timm's fused qkv weight
    k = flax["Transformer/encoderblock_3/MultiHeadDotProductAttention_1/key/kernel"]
    assert np.array_equal(k[5, 7, 9], sd["blocks.3.attn.qkv.weight"][768 + 7 * 64 + 9, 5].numpy())
    path = tmp_path / "B_16.npz"
    np.savez(path, **flax)
    monkeypatch.setenv("SSL4GIE_AUGREG_NPZ", str(path))
    m2 = models.VisionTransformer_from_Any(False, 0, False, None, False, None, 768, 12, 12, "cls", ImageNet_weights=True)
    got = m2.state_dict()
    for k, v in sd.items():
        assert torch.equal(got[k], v), k
    monkeypatch.delenv("SSL4GIE_AUGREG_NPZ")
    with pytest.raises(RuntimeError, match="no network"):
        models.VisionTransformer_from_Any(False, 0, False, None, False, None, 768, 12, 12, "cls", ImageNet_weights=True)


def test_position_table_resize_matches_f_interpolate():
    """Models/models.py `_interp_pos` (two small matmuls with the separable align_corners=True
    weights) against the reference's formulation (models.py:310-323: F.interpolate, bilinear,
    align_corners=True) on the CPU, values and the table gradient."""
    import torch
    import torch.nn.functional as F
    from ssl4gie_amd.Models.models import _interp_pos
    D = 12
    g0 = torch.Generator().manual_seed(3)
    for g in (14, 16, 37, 64, 7):
        pos = torch.randn(1, 197, D, generator=g0, requires_grad=True)
        ref = F.interpolate(pos[:, 1:, :].transpose(1, 2).reshape(1, D, 14, 14), size=(g, g), mode="bilinear",
                            align_corners=True).reshape(1, D, g * g).transpose(1, 2)
        got = _interp_pos(pos, g, D)
        assert (got - ref).abs().max() < 2e-6
        w = torch.randn(1, g * g, D, generator=g0)
        (gr,) = torch.autograd.grad(ref, pos, w, retain_graph=True)
        (gg,) = torch.autograd.grad(got, pos, w)
        assert (gr - gg).abs().max() < 2e-5 * max(1.0, float(gr.abs().max()))
        assert gg[0, 0].abs().max() == 0  # the cls row gets no gradient


@pytest.mark.timeout(400)
def test_bench_self_launches_its_ranks_when_invoked_plainly():
    """`python bench.py --gpus 2 ...` with no torchrun environment (how the driver invokes N = 1) must
    start its own ranks: the parent spawns `torch.distributed.run` children before touching the GPU,
    relays rank 0's line and exits with the launcher's code (it used to die on an assert)"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                       env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["rendezvous"] == 2 and line["rank_sum"] == 1.0
    # the N > 1 line proves itself: the backend saw both ranks, the replicas hold the same
    # parameters after the steps, the exposed communication time is reported
    dp = line["dp"]
    assert dp["backend_world"] == 2 and dp["replicas_in_sync"] is True and dp["backend"] == "gloo"
    assert dp["param_checksum"][0] == dp["param_checksum"][1]
    assert isinstance(dp["exposed_comm_ms"], float) and dp["grad_collectives_per_step"] >= 1
    # ... and a run whose replicas drifted apart fails loudly on every rank
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                       env=dict(env, SSL4GIE_BENCH_DESYNC_RANK="1"), capture_output=True, text=True, timeout=170)
    assert r.returncode != 0 and "self-check FAILED" in r.stderr
    bad = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert bad["dp"]["replicas_in_sync"] is False
    # a failing rank is reported through the exit code
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only",
                        "--no-such-flag"], env=env, capture_output=True, text=True, timeout=170)
    assert r.returncode != 0
