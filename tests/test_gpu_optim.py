"""GPU parity of the arena optimizer kernels against torch.optim.AdamW and the
host-side LARS mirror of the reference (`ssl4gie_amd.Models.moco_v3.moco.optimizer.LARS`, itself
pinned by the G8 fixture of the reference's own class)."""
import copy

import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from ssl4gie_amd import _lib
    _lib.load()


def _toy():
    from ssl4gie_amd.engine import EngineModule

    class Toy(EngineModule):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Linear(40, 72)
            self.b = torch.nn.Linear(72, 24)
            self.norm = torch.nn.LayerNorm(24)
            self.frozen = torch.nn.Parameter(torch.randn(3, 5), requires_grad=False)
            self.unused = torch.nn.Parameter(torch.randn(7))  # never receives a gradient
            self.conv = torch.nn.Conv2d(8, 16, 3)
    torch.manual_seed(0)
    return Toy().to(DEV)


def _fake_grads(m, seed):
    g = torch.Generator("cpu").manual_seed(seed)
    for name, p in m.named_parameters():
        if p.requires_grad and name != "unused":
            p.grad = torch.randn(p.shape, generator=g).to(DEV)
        else:
            p.grad = None


def _groups(m):
    decay = [p for n, p in m.named_parameters() if p.requires_grad and p.ndim > 1]
    no_decay = [p for n, p in m.named_parameters() if p.requires_grad and p.ndim <= 1]
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": 0.05}]


def test_arena_adamw_matches_torch_adamw():
    from ssl4gie_amd.optim import ArenaAdamW
    m = _toy()
    ref = copy.deepcopy(m)
    m.arena()
    opt = ArenaAdamW(m, _groups(m), lr=1e-2, betas=(0.9, 0.95))
    topt = torch.optim.AdamW(_groups(ref), lr=1e-2, betas=(0.9, 0.95))
    frozen0, unused0 = m.frozen.detach().clone(), m.unused.detach().clone()
    for step in range(4):
        _fake_grads(m, 10 + step)
        for (n1, p1), (n2, p2) in zip(m.named_parameters(), ref.named_parameters()):
            p2.grad = None if p1.grad is None else p1.grad.clone()
        # the gradients were produced outside the arena (plain tensors): step() adopts them
        if step == 2:  # a learning-rate schedule edits param_groups in place
            for g in opt.param_groups + topt.param_groups:
                g["lr"] = 5e-3
        opt.step()
        topt.step()
    for (n1, p1), (n2, p2) in zip(m.named_parameters(), ref.named_parameters()):
        assert rel_err(p1, p2) < 1e-5, n1
    assert torch.equal(m.frozen, frozen0) and torch.equal(m.unused, unused0)


def test_arena_lars_matches_reference_lars():
    from ssl4gie_amd.Models.moco_v3.moco.optimizer import LARS
    from ssl4gie_amd.optim import ArenaLARS
    m = _toy()
    ref = copy.deepcopy(m)
    m.arena()
    ps = [p for p in m.parameters() if p.requires_grad]
    opt = ArenaLARS(m, ps, lr=0.3, weight_decay=1e-2, momentum=0.9)
    topt = LARS([p for p in ref.parameters() if p.requires_grad], lr=0.3, weight_decay=1e-2, momentum=0.9)
    for step in range(3):
        _fake_grads(m, 20 + step)
        for (n1, p1), (n2, p2) in zip(m.named_parameters(), ref.named_parameters()):
            p2.grad = None if p1.grad is None else p1.grad.clone()
        a = m.arena()
        for p in m.parameters():
            if p.grad is not None:
                v = a.grad_view(p)
                v.copy_(p.grad)
                p.grad = v
        opt.step()
        topt.step()
    for (n1, p1), (n2, p2) in zip(m.named_parameters(), ref.named_parameters()):
        assert rel_err(p1, p2) < 1e-5, n1


def test_arena_adamw_trains_mae_and_refreshes_operand_caches():
    """end to end: the fused step must invalidate the bf16 weight copies (weights_epoch)"""
    from ssl4gie_amd.Models.mae import models_mae
    from ssl4gie_amd.optim import ArenaAdamW
    torch.manual_seed(0)
    m = models_mae.MaskedAutoencoderViT(embed_dim=192, depth=2, num_heads=3, decoder_embed_dim=128,
                                        decoder_depth=1, decoder_num_heads=4, norm_pix_loss=True).to(DEV)
    m.set_precision("bf16")
    imgs = torch.randn(16, 3, 224, 224, generator=torch.Generator().manual_seed(1)).to(DEV)
    noise = torch.rand(16, 196, generator=torch.Generator().manual_seed(2)).to(DEV)
    opt = ArenaAdamW(m, [p for p in m.parameters() if p.requires_grad], lr=2e-3, betas=(0.9, 0.95),
                     weight_decay=0.05)
    losses = []
    for _ in range(8):
        opt.zero_grad()
        loss, _, _ = m(imgs, mask_ratio=0.75, noise=noise)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < 0.9 * losses[0], losses


def test_arena_optimizer_state_dict_round_trip():
    """resume (reference `save_model` / `main_moco.py:310-316` store optimizer.state_dict()): a
    restored ArenaAdamW / ArenaLARS continues bit-identically"""
    from ssl4gie_amd.optim import ArenaAdamW, ArenaLARS
    for cls, kw in ((ArenaAdamW, dict(lr=1e-2, betas=(0.9, 0.95))), (ArenaLARS, dict(lr=0.3, weight_decay=1e-2))):
        m1 = _toy()
        m2 = copy.deepcopy(m1)
        m1.arena(), m2.arena()
        mk = lambda m: cls(m, _groups(m) if cls is ArenaAdamW else [p for p in m.parameters() if p.requires_grad], **kw)
        o1 = mk(m1)
        for step in range(2):
            _fake_grads(m1, 30 + step)
            o1.step()
        sd = o1.state_dict()
        assert sd["step_count"] == 2 and all(v is not None for v in sd["state"].values())
        m2.load_state_dict(m1.state_dict())
        o2 = mk(m2)
        o2.load_state_dict(sd)
        for g in o1.param_groups + o2.param_groups:
            g["lr"] = g["lr"] * 0.5
        for o, m in ((o1, m1), (o2, m2)):
            _fake_grads(m, 40)
            o.step()
        for (n1, p1), (n2, p2) in zip(m1.named_parameters(), m2.named_parameters()):
            assert torch.equal(p1, p2), n1
        a = m1.arena()
        assert all(p.grad is None or p.grad.data_ptr() == a.grad_view(p).data_ptr() for p in m1.parameters())


def test_bf16_operand_copies_follow_fused_and_plain_optimizer_steps():
    """torch's fused AdamW does not bump Tensor._version: the operand caches must still see every
    optimizer step (engine registers a global optimizer post-hook), and the arena-wide refresh (one
    flat cast + one batched transposed cast) must equal the per-parameter casts, including after an
    in-place edit of a single parameter."""
    from ssl4gie_amd.Models.mae import models_mae
    torch.manual_seed(0)
    model = models_mae.MaskedAutoencoderViT(img_size=32, patch_size=8, embed_dim=64, depth=2, num_heads=2,
                                            decoder_embed_dim=64, decoder_depth=1, decoder_num_heads=2,
                                            mlp_ratio=3).to(DEV).set_precision("bf16")
    imgs = torch.randn(4, 3, 32, 32, device=DEV)
    for fused in (True, False):
        opt = torch.optim.AdamW(model.parameters(), lr=1e-2, fused=fused)
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            loss, _, _ = model(imgs, mask_ratio=0.5)
            loss.backward()
            opt.step()
        loss, _, _ = model(imgs, mask_ratio=0.5)  # refreshes the caches
        for name, p in model.named_parameters():
            if p.ndim == 2 and "pos_embed" not in name:
                w, wt = model.lp_cache.get(p, torch.bfloat16)
                ref = p.detach().to(torch.bfloat16)
                assert torch.equal(w, ref), (fused, name)
                assert torch.equal(wt, ref.t().contiguous()), (fused, name)
    w0 = model.blocks[0].mlp.fc1.weight
    with torch.no_grad():
        w0.mul_(0.5)  # version moves, epoch does not
    w, wt = model.lp_cache.get(w0, torch.bfloat16)
    ref = w0.detach().to(torch.bfloat16)
    assert torch.equal(w, ref) and torch.equal(wt, ref.t().contiguous())


def test_batched_transposed_cast_ragged_matrices():
    """ssl4gie_cast_transpose_batch on matrices that are not multiples of its 64 x 64 tile (and not of
    4: the scalar path), against per-matrix torch transposes; bit-exact (same RNE conversion)."""
    from ssl4gie_amd import _lib, ops
    shapes = [(64, 64), (192, 64), (7, 5), (130, 257), (1, 300), (100, 4), (33, 64)]
    offs, total = [], 0
    for r, c in shapes:
        offs.append(total)
        total += (r * c + 63) // 64 * 64
    src = torch.randn(total, device=DEV)
    dst = torch.zeros(total, dtype=torch.bfloat16, device=DEV)
    starts = [0]
    for r, c in shapes:
        starts.append(starts[-1] + ((r + 63) // 64) * ((c + 63) // 64))
    t = lambda v, dt: torch.tensor(v, dtype=dt, device=DEV)
    off_t, r_t, c_t, s_t = t(offs, torch.int64), t([s[0] for s in shapes], torch.int32), \
        t([s[1] for s in shapes], torch.int32), t(starts, torch.int32)
    _lib.check(_lib.load().ssl4gie_cast_transpose_batch(ops.ptr(src), ops.ptr(dst), ops.ptr(off_t), ops.ptr(r_t),
                                                        ops.ptr(c_t), ops.ptr(s_t), len(shapes), starts[-1],
                                                        ops.stream()), "cast_transpose_batch")
    for (r, c), o in zip(shapes, offs):
        ref = src[o:o + r * c].view(r, c).to(torch.bfloat16).t().contiguous()
        assert torch.equal(dst[o:o + r * c].view(c, r), ref), (r, c)


def test_arena_adamw_in_backward_updates_match_the_plain_step():
    """overlap_backward=True enqueues each block's update behind that block's backward on a side
    stream; the arithmetic is the same, so parameters, moments and the bf16 operand copies must be
    bitwise those of the plain step — over several steps, with a learning-rate change in between"""
    from ssl4gie_amd.Models.mae import models_mae
    from ssl4gie_amd.optim import ArenaAdamW

    def run(overlap):
        torch.manual_seed(0)
        model = models_mae.MaskedAutoencoderViT(img_size=32, patch_size=8, embed_dim=64, depth=3, num_heads=2,
                                                decoder_embed_dim=64, decoder_depth=2, decoder_num_heads=2,
                                                mlp_ratio=4).to(DEV).set_precision("bf16")
        decay = [p for p in model.parameters() if p.requires_grad and p.ndim > 1]
        rest = [p for p in model.parameters() if p.requires_grad and p.ndim <= 1]
        opt = ArenaAdamW(model, [{"params": decay, "weight_decay": 0.05}, {"params": rest, "weight_decay": 0.0}],
                         lr=1e-2, betas=(0.9, 0.95), overlap_backward=overlap)
        imgs = torch.randn(8, 3, 32, 32, generator=torch.Generator().manual_seed(3)).to(DEV)
        noise = torch.rand(8, 16, generator=torch.Generator().manual_seed(4)).to(DEV)
        for it in range(5):
            if it == 3:
                for g in opt.param_groups:
                    g["lr"] = 3e-3
            opt.zero_grad(set_to_none=True)
            loss, _, _ = model(imgs, mask_ratio=0.5, noise=noise)
            loss.backward()
            opt.step()
        torch.cuda.synchronize()
        used = len(opt._done) == 0 and opt._overlap
        return model, opt, float(loss), used

    torch.manual_seed(1)
    m0, o0, l0, _ = run(False)
    m1, o1, l1, active = run(True)
    assert active, "the overlapped mode fell back to the plain step"
    assert l0 == l1
    for (n0, p0), (_, p1) in zip(m0.named_parameters(), m1.named_parameters()):
        assert torch.equal(p0, p1), n0
    assert torch.equal(o0.exp_avg, o1.exp_avg) and torch.equal(o0.exp_avg_sq, o1.exp_avg_sq)
    w = m1.blocks[1].mlp.fc2.weight
    lp, lpt = m1.lp_cache.get(w, torch.bfloat16)
    ref = w.detach().to(torch.bfloat16)
    assert torch.equal(lp, ref) and torch.equal(lpt, ref.t().contiguous())


def test_bf16_training_curve_is_the_same_under_fused_and_foreach_adamw():
    """guards the whole chain optimizer step -> operand-copy refresh -> next forward: with stale bf16
    weights (what torch's fused AdamW used to cause) the two curves part after the first step"""
    from ssl4gie_amd.Models.mae import models_mae
    curves = []
    for fused in (True, False):
        torch.manual_seed(0)
        model = models_mae.MaskedAutoencoderViT(img_size=32, patch_size=8, embed_dim=64, depth=2, num_heads=2,
                                                decoder_embed_dim=64, decoder_depth=1, decoder_num_heads=2,
                                                mlp_ratio=4).to(DEV).set_precision("bf16")
        opt = torch.optim.AdamW(model.parameters(), lr=3e-3, betas=(0.9, 0.95), fused=fused)
        imgs = torch.randn(16, 3, 32, 32, generator=torch.Generator().manual_seed(7)).to(DEV)
        noise = torch.rand(16, 16, generator=torch.Generator().manual_seed(8)).to(DEV)
        losses = []
        for _ in range(12):
            opt.zero_grad(set_to_none=True)
            loss, _, _ = model(imgs, mask_ratio=0.5, noise=noise)
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        curves.append(losses)
    a, b = curves
    assert a[-1] < 0.8 * a[0], a  # it trains
    assert max(abs(x - y) for x, y in zip(a, b)) < 2e-2 * a[0], (a, b)


def test_operand_copies_after_arena_step_then_load_state_dict():
    """ADVICE r2 (engine.py:117): ArenaAdamW.step() marks the flat bf16 shadow current for the new
    epoch; a load_state_dict() before the next forward moves only Tensor._version.  The refresh must
    not skip the flat cast then (forward would use the optimizer's W, the data gradient the loaded
    W^T): every operand copy equals the loaded weights and the loss equals a fresh model's."""
    from ssl4gie_amd.Models.mae import models_mae
    from ssl4gie_amd.optim import ArenaAdamW

    def make():
        torch.manual_seed(3)
        return models_mae.MaskedAutoencoderViT(img_size=32, patch_size=8, embed_dim=64, depth=2, num_heads=2,
                                               decoder_embed_dim=64, decoder_depth=1, decoder_num_heads=2,
                                               mlp_ratio=3).to(DEV).set_precision("bf16")
    model = make()
    imgs = torch.randn(4, 3, 32, 32, device=DEV)
    noise = torch.rand(4, 16, device=DEV)
    opt = ArenaAdamW(model, [p for p in model.parameters() if p.requires_grad], lr=5e-2)
    for _ in range(2):
        opt.zero_grad()
        loss, _, _ = model(imgs, mask_ratio=0.5, noise=noise)
        loss.backward()
        opt.step()                       # writes the flat shadow itself, bumps the epoch
    fresh = make()
    sd = {k: v.clone() for k, v in fresh.state_dict().items()}
    model.load_state_dict(sd)            # versions move, the epoch does not
    loss, _, _ = model(imgs, mask_ratio=0.5, noise=noise)
    ref, _, _ = fresh(imgs, mask_ratio=0.5, noise=noise)
    assert float(loss) == float(ref)
    for name, p in model.named_parameters():
        if p.ndim == 2 and "pos_embed" not in name:
            w, wt = model.lp_cache.get(p, torch.bfloat16)
            r = p.detach().to(torch.bfloat16)
            assert torch.equal(w, r) and torch.equal(wt, r.t().contiguous()), name
