"""GPU: the reference's own training-loop STATEMENT SEQUENCES run unchanged on engine models wrapped in
ssl4gie_amd.parallel.DataParallel — two processes on the one device (gloo between them), different data
per rank, no finish() call anywhere:

  * Depth_estimation/train_depth.py:35-48   model.train(); optimizer.zero_grad();
        with torch.cuda.amp.autocast(): output = model(data); loss = loss_fn(output, target)
        scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update();
        dist.all_reduce(loss); loss /= world_size
  * Models/mae/engine_pretrain.py:39-69 with Models/mae/util/misc.py:251-271
        (NativeScalerWithGradNormCount: scale -> backward -> unscale_ -> grad norm -> step -> update),
        accum_iter = 2 (two backward passes per optimizer step, as the reference accumulates: no no_sync)

Each loop is run twice from the same weights — once verbatim (autocast + GradScaler) and once with a plain
`loss.backward(); optimizer.step()` — and the loss sequences must agree: those lines really are no-ops for
the engine.  After the loops every rank holds the same weights, and the MAE gradient
of the two ranks equals a single process's gradient on the concatenated batch."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _same_on_all_ranks(model, dist):
    flat = torch.cat([p.detach().flatten() for p in model.parameters()]).double()
    sig = torch.stack([flat.sum(), flat.abs().sum(), (flat * flat).sum()]).cpu()
    sigs = [torch.empty_like(sig) for _ in range(dist.get_world_size())]
    dist.all_gather(sigs, sig)
    return all(torch.equal(s, sigs[0]) for s in sigs)


def _depth_loop(rank, world, verbatim, steps=4):
    """train_depth.py:22-78 (train_epoch) + :226-229,280 (DDP construction, AdamW, loss)"""
    import torch.distributed as dist
    from ssl4gie_amd import utils
    from ssl4gie_amd.losses import ScaleAndShiftInvariantLoss
    from ssl4gie_amd.parallel import DataParallel
    torch.manual_seed(11)
    model = utils.get_MAE_backbone(None, False, 1, False, "depth", False)
    model.cuda(0).set_precision("fp32")
    model = DataParallel(model, device_ids=[0], find_unused_parameters=True)
    optimizer = torch.optim.AdamW(model.parameters(), lr=1e-4)
    loss_fn = ScaleAndShiftInvariantLoss()
    scaler = torch.cuda.amp.GradScaler()
    g = torch.Generator().manual_seed(300 + rank)
    loader = []
    for _ in range(steps):
        data = torch.randn(2, 3, 224, 224, generator=g)
        target = torch.rand(2, 1, 224, 224, generator=g)
        target = torch.where(torch.rand(2, 1, 224, 224, generator=g) < 0.1, torch.zeros(()), target)
        loader.append((data, target))
    losses = []
    model.train()
    for batch_idx, (data, target) in enumerate(loader):
        data, target = data.cuda(0), target.cuda(0)
        optimizer.zero_grad()
        if verbatim:
            with torch.cuda.amp.autocast():
                output = model(data)
                loss = loss_fn(output, target)
            scaler.scale(loss).backward()
            scaler.step(optimizer)
            scaler.update()
        else:
            output = model(data)
            loss = loss_fn(output, target)
            loss.backward()
            optimizer.step()
        loss = loss.detach().float()
        dist.all_reduce(loss)
        loss /= world
        losses.append(loss.item())
        dist.barrier()
    model.eval()
    assert not model.module.training
    info = dict(passes=model.n_passes, overlapped=model.n_overlapped, collectives=model.n_collectives,
                late=model.n_late, same=_same_on_all_ranks(model, dist),
                prefixed=all(k.startswith("module.") for k in model.state_dict()),
                plain=not any(k.startswith("module.") for k in model.module.state_dict()))
    return losses, info


def _mae_loop(rank, world, verbatim, steps=6, accum_iter=2, arena=False):
    """engine_pretrain.py:39-69 with misc.NativeScalerWithGradNormCount (misc.py:251-271)"""
    import torch.distributed as dist
    from functools import partial
    from oracle import mae_ref, synth
    from ssl4gie_amd.Models.mae.models_mae import MaskedAutoencoderViT
    from ssl4gie_amd.parallel import DataParallel
    cfg = mae_ref.MAEConfig(**{**mae_ref.TINY.__dict__, "norm_pix_loss": True})
    model = MaskedAutoencoderViT(img_size=cfg.img_size, patch_size=cfg.patch_size, in_chans=cfg.in_chans,
                                 embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads,
                                 decoder_embed_dim=cfg.decoder_embed_dim, decoder_depth=cfg.decoder_depth,
                                 decoder_num_heads=cfg.decoder_num_heads, mlp_ratio=cfg.mlp_ratio,
                                 norm_layer=partial(torch.nn.LayerNorm, eps=cfg.ln_eps), norm_pix_loss=True)
    model.load_state_dict(synth.mae_state_dict(cfg, 1))
    model.cuda(0).set_precision("fp32")
    model_without_ddp = model
    model = DataParallel(model, device_ids=[0], find_unused_parameters=True)   # main_pretrain.py:175
    if arena:   # the bench's optimizer behind the same GradScaler statements (duck-typed param_groups / step)
        from ssl4gie_amd.optim import ArenaAdamW
        optimizer = ArenaAdamW(model_without_ddp, [p for p in model_without_ddp.parameters() if p.requires_grad],
                               lr=1.5e-4, betas=(0.9, 0.95))
    else:
        optimizer = torch.optim.AdamW(model_without_ddp.parameters(), lr=1.5e-4, betas=(0.9, 0.95))
    _scaler = torch.cuda.amp.GradScaler()

    def loss_scaler(loss, optimizer, parameters=None, update_grad=True):   # misc.py:257-271
        _scaler.scale(loss).backward(create_graph=False)
        if update_grad:
            _scaler.unscale_(optimizer)
            norm = torch.norm(torch.stack([torch.norm(p.grad.detach(), 2.0) for p in parameters if p.grad is not None]), 2.0)
            _scaler.step(optimizer)
            _scaler.update()
        else:
            norm = None
        return norm

    losses, norms = [], []
    model.train(True)
    optimizer.zero_grad()
    for data_iter_step in range(steps):
        samples = synth.synth_images(4, cfg, seed=10 * rank + data_iter_step).cuda(0, non_blocking=True)
        noise = torch.from_numpy(synth.synth_noise(4, cfg.num_patches, seed=50 * rank + data_iter_step)).cuda(0)
        if verbatim:
            with torch.cuda.amp.autocast():
                loss, _, _ = model(samples, mask_ratio=0.75, noise=noise)
        else:
            loss, _, _ = model(samples, mask_ratio=0.75, noise=noise)
        loss_value = loss.item()
        assert np.isfinite(loss_value)
        loss /= accum_iter
        update = (data_iter_step + 1) % accum_iter == 0
        if verbatim:
            norm = loss_scaler(loss, optimizer, parameters=model.parameters(), update_grad=update)
        else:
            loss.backward()
            norm = None
            if update:
                norm = torch.norm(torch.stack([torch.norm(p.grad.detach(), 2.0) for p in model.parameters()
                                               if p.grad is not None]), 2.0)
                optimizer.step()
        if update:
            optimizer.zero_grad()
            norms.append(float(norm))
        torch.cuda.synchronize()
        losses.append(loss_value)
    info = dict(passes=model.n_passes, same=_same_on_all_ranks(model, dist), late=model.n_late)
    return losses, norms, info


def _mae_grad_two_ranks_vs_one(rank, world):
    """gradient after ONE wrapped backward on rank-specific data == single-process gradient on the
    concatenated batch (the MAE loss is a mean over 147 masked patches per image: mean of the ranks'
    losses == loss of the concatenation)"""
    from functools import partial
    from oracle import mae_ref, synth
    from ssl4gie_amd.Models.mae.models_mae import MaskedAutoencoderViT
    from ssl4gie_amd.parallel import DataParallel
    cfg = mae_ref.MAEConfig(**{**mae_ref.TINY.__dict__, "norm_pix_loss": True})

    def make():
        m = MaskedAutoencoderViT(img_size=cfg.img_size, patch_size=cfg.patch_size, in_chans=cfg.in_chans,
                                 embed_dim=cfg.embed_dim, depth=cfg.depth, num_heads=cfg.num_heads,
                                 decoder_embed_dim=cfg.decoder_embed_dim, decoder_depth=cfg.decoder_depth,
                                 decoder_num_heads=cfg.decoder_num_heads, mlp_ratio=cfg.mlp_ratio,
                                 norm_layer=partial(torch.nn.LayerNorm, eps=cfg.ln_eps), norm_pix_loss=True)
        m.load_state_dict(synth.mae_state_dict(cfg, 1))
        return m.cuda(0).set_precision("fp32")

    imgs = [synth.synth_images(4, cfg, seed=70 + r) for r in range(world)]
    noise = [torch.from_numpy(synth.synth_noise(4, cfg.num_patches, seed=80 + r)) for r in range(world)]
    one = make()
    loss1, _, _ = one(torch.cat(imgs).cuda(0), noise=torch.cat(noise).cuda(0))
    loss1.backward()
    ddp = DataParallel(make(), bucket_bytes=1 << 16)   # small buckets: several slices leave in backward
    worst = 0.0
    for _ in range(2):   # the second pass runs with the learnt schedule (slices leave during backward)
        for p in ddp.parameters():
            p.grad = None
        loss, _, _ = ddp(imgs[rank].cuda(0), noise=noise[rank].cuda(0))
        loss.backward()
        torch.cuda.synchronize()
        for (n, p), q in zip(ddp.module.named_parameters(), one.parameters()):
            if q.grad is None:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
                continue
            den = float(q.grad.abs().max()) or 1.0
            worst = max(worst, float((p.grad - q.grad).abs().max()) / den)
    return worst, ddp.n_overlapped


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SSL4GIE_COMM_CUS="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    out = {}
    try:
        out["depth_verbatim"] = _depth_loop(rank, world, True)
        out["depth_plain"] = _depth_loop(rank, world, False)
        out["mae_verbatim"] = _mae_loop(rank, world, True)
        out["mae_plain"] = _mae_loop(rank, world, False)
        out["mae_arena"] = _mae_loop(rank, world, True, arena=True)
        out["mae_grad"] = _mae_grad_two_ranks_vs_one(rank, world)
    except Exception as e:  # noqa: BLE001 - reported to the parent
        import traceback
        out["error"] = traceback.format_exc()
    q.put((rank, out))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_reference_statement_sequences_two_ranks_one_device():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=800) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for r in (0, 1):
        assert "error" not in res[r], res[r]["error"]
    for r in (0, 1):
        lv, iv = res[r]["depth_verbatim"]
        lp, ip = res[r]["depth_plain"]
        assert np.allclose(lv, lp, rtol=1e-4), (lv, lp)        # autocast + GradScaler: no-ops for the engine
        for info in (iv, ip):
            assert info["passes"] == 4 and info["same"] and info["prefixed"] and info["plain"], info
            assert info["overlapped"] >= 3, info                  # slices left while backward was running
        mv, nv, jv = res[r]["mae_verbatim"]
        mp_, np_, jp = res[r]["mae_plain"]
        assert np.allclose(mv, mp_, rtol=1e-4), (mv, mp_)
        assert np.allclose(nv, np_, rtol=1e-3), (nv, np_)
        assert jv["passes"] == 6 and jp["passes"] == 6 and jv["same"] and jp["same"]
        ma, na, ja = res[r]["mae_arena"]     # ArenaAdamW under the scaler == torch.optim.AdamW under the scaler
        assert np.allclose(ma, mv, rtol=1e-4) and np.allclose(na, nv, rtol=1e-3) and ja["passes"] == 6 and ja["same"]
        worst, overlapped = res[r]["mae_grad"]
        assert worst < 1e-4, worst
        assert overlapped >= 1
    # the logged loss is the mean over ranks: identical on both
    assert np.allclose(res[0]["depth_verbatim"][0], res[1]["depth_verbatim"][0], rtol=1e-6)
