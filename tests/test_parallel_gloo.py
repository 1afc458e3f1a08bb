"""CPU, world_size 2 over gloo: the bucketed arena all-reduce of ssl4gie_amd.parallel averages
gradients exactly like DDP (reference train_depth.py:226-229) and overlap bookkeeping is sound."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _SinkLinearFn(torch.autograd.Function):
    """CPU stand-in for an engine node: y = x W^T + b whose backward asks the model's GradSink where
    to put the parameter gradients (arena slice, first use writes / later uses accumulate) exactly
    like engine.LinearFn does — only the arithmetic is torch instead of the HIP library."""

    @staticmethod
    def forward(ctx, x, w, b, sink):
        ctx.save_for_backward(x, w, b)
        ctx.sink = sink
        return x @ w.t() + b

    @staticmethod
    def backward(ctx, dy):
        x, w, b = ctx.saved_tensors
        (tw, tb), acc, rets = ctx.sink.plan([w, b])
        dw, db = dy.reshape(-1, dy.shape[-1]).t() @ x.reshape(-1, x.shape[-1]), dy.reshape(-1, dy.shape[-1]).sum(0)
        for t, g in ((tw, dw), (tb, db)):
            if t is not None:
                t.add_(g) if acc else t.copy_(g)
        ctx.sink.note_done([w, b])  # what the block executor reports per transformer block
        return dy @ w, rets[0], rets[1], None


class _Toy(nn.Module):
    """Stands in for an EngineModule tree: parameters in registration order in ONE arena, an adopted
    child (`head`, like the DPT decoder inside ViT_from_MAE) that shares the parent's sink, a frozen
    twin (`momentum`, like MoCo's momentum encoder) in the middle of the arena, a parameter nothing
    uses (`unused`, like `norm.*` in dense mode) and one whose gradient comes from plain torch ops
    (`pos`, like the interpolated pos_embed of the detection trunk)."""

    def __init__(self, width=32, depth=6, frozen=0):
        super().__init__()
        self.pos = nn.Parameter(torch.zeros(1, width))
        self.stem = nn.Linear(8, width)
        self.blocks = nn.ModuleList([nn.Linear(width, width) for _ in range(depth)])
        self.unused = nn.Linear(width, width)
        if frozen:
            self.momentum = nn.Linear(frozen, frozen)
            for q in self.momentum.parameters():
                q.requires_grad = False
        self.head = nn.Module()
        self.head.fc = nn.Linear(width, 4)
        self._a = None
        self._s = None

    def arena(self):
        from ssl4gie_amd.engine import GradSink, ParamArena
        if self._a is None:
            self._a = ParamArena(list(self.parameters()))
            self._s = GradSink(self._a)
        return self._a

    def sink(self):
        self.arena()
        return self._s

    def _lin(self, x, lin):
        return _SinkLinearFn.apply(x, lin.weight, lin.bias, self._s)

    def trunk(self, x):
        x = self._lin(x, self.stem) + self.pos * 2.0   # torch-op gradient for `pos`
        for blk in self.blocks:
            x = torch.tanh(self._lin(x, blk))
        return self._lin(x, self.head.fc)

    def forward(self, x1, x2=None):
        self.arena()
        self._s.new_pass()
        y = self.trunk(x1)
        if x2 is not None:  # two views through the SAME parameters (MoCo / Barlow Twins)
            y = y + 0.5 * self.trunk(x2)
        return (y ** 2).mean()


def _single_process_grads(seed_model, batches, **kw):
    """gradient of the mean loss over ranks, computed in one process with plain torch parameters"""
    torch.manual_seed(seed_model)
    m = _Toy(**kw)
    total = 0
    for b in batches:
        m2 = m
        x = b[0]
        def lin(t, l):
            return t @ l.weight.t() + l.bias
        def trunk(t):
            t = lin(t, m2.stem) + m2.pos * 2.0
            for blk in m2.blocks:
                t = torch.tanh(lin(t, blk))
            return lin(t, m2.head.fc)
        y = trunk(b[0])
        if len(b) > 1:
            y = y + 0.5 * trunk(b[1])
        total = total + (y ** 2).mean() / len(batches)
    total.backward()
    return {k: (p.grad.clone() if p.grad is not None else None) for k, p in m.named_parameters()}


def _batches(world, two_view):
    out = []
    for r in range(world):
        g = torch.Generator().manual_seed(500 + r)
        b = [torch.randn(5, 8, generator=g)]
        if two_view:
            b.append(torch.randn(5, 8, generator=g))
        out.append(b)
    return out


def _worker(rank, world, port, q, two_view, frozen, bucket_bytes):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ssl4gie_amd.parallel import DataParallel
    torch.manual_seed(1234 + rank)  # different initial weights per rank: broadcast must fix it
    m = _Toy(frozen=frozen)
    torch.manual_seed(1234)
    ref_init = _Toy(frozen=frozen)  # rank 0's initial weights
    # the reference's constructor call, class name replaced (train_depth.py:226-229)
    model = DataParallel(m, device_ids=[0], find_unused_parameters=True, bucket_bytes=bucket_bytes)
    a = m.arena()
    same_weights = all(torch.equal(p, r) for p, r in zip(m.parameters(), ref_init.parameters()))
    batches = _batches(world, two_view)  # DIFFERENT data per rank
    expect = _single_process_grads(1234, batches, frozen=frozen)
    opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.0)
    res = []
    for step in range(3):  # step 0 learns the unused set; steps 1, 2 overlap fully
        model.train()
        opt.zero_grad(set_to_none=True)
        before, before_ov = model.n_collectives, model.n_overlapped
        loss = model(*batches[rank])
        loss.backward()             # the exchange finishes in here: no finish() anywhere
        during = model.n_overlapped - before_ov
        ok = model.n_passes == step + 1
        for k, p in m.named_parameters():
            e = expect[k]
            if e is None:
                ok &= p.grad is None or float(p.grad.abs().max()) == 0.0
            else:
                ok &= p.grad is not None and torch.allclose(p.grad, e, rtol=1e-5, atol=1e-7)
                ok &= p.grad.data_ptr() == a.grad_view(p).data_ptr()  # incl. the adopted torch-op gradient
        opt.step()
        model.finish()              # idempotent: nothing left to do
        ok &= model.n_passes == step + 1
        res.append((bool(ok), during, model.n_collectives - before, model.n_late))
    loss_mean = model.all_reduce_mean(torch.tensor([float(rank + 1)]))
    keys_ok = all(k.startswith("module.") for k in model.state_dict()) and \
        set(model.module.state_dict()) == {k[len("module."):] for k in model.state_dict()}
    q.put((rank, same_weights and keys_ok, res, float(loss_mean)))
    dist.destroy_process_group()


def _run(two_view, frozen, bucket_bytes, target=None, extra=()):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target or _worker, args=(r, 2, port, q) + ((two_view, frozen, bucket_bytes) if target is None else tuple(extra)))
             for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    return res


@pytest.mark.timeout(120)
@pytest.mark.parametrize("two_view", [False, True])
def test_bucketed_allreduce_world2(two_view):
    """gradient mean over 2 ranks (different data per rank) == single-process gradient of the mean
    loss, with NO finish() call: the pass is closed from inside backward(), buckets go out DURING
    backward — also when every parameter is used twice per step (the slice of a parameter must not
    go out before its second use has accumulated into it), for the adopted child's and the
    torch-op-produced gradients, and around a parameter that never receives a gradient"""
    for rank, same_w, steps, lm in _run(two_view, 0, 4 * 2000):
        assert same_w, "rank-0 parameter broadcast / module.-prefixed state_dict"
        for ok, during, total, late in steps:
            assert ok, "gradient average"
            assert late == 0
        # step 0: the never-used parameter sits above the blocks and holds the frontier until the
        # end of backward; afterwards it is known and most slices leave while backward is still running
        assert steps[1][1] >= 2 and steps[2][1] >= 2, steps
        assert steps[1][2] <= steps[1][1] + 3   # (+ the tail slice of the learnt-unused run)
        assert abs(lm - 1.5) < 1e-6


@pytest.mark.timeout(120)
def test_frozen_run_is_not_communicated_world2():
    """a large frozen block in the middle of the arena (MoCo's momentum encoder) splits it into
    segments: trainable slices either side are averaged, the frozen run is never sent; once the unused
    parameter is known, its run travels as one small tail slice at the end of every pass (so that a rank
    whose graph does use it is still averaged correctly)"""
    for rank, same_w, steps, lm in _run(True, 600, 1 << 30):
        assert same_w
        for i, (ok, during, total, late) in enumerate(steps):
            assert ok and late == 0
            assert total == (2 if i == 0 else 3), "one slice per trainable segment (bucket larger than either) + the tail"


def _probe_worker(rank, world, port, q):
    """linear probe / frozen finetune (reference `frozen=True`, Models/models.py:138-142,341-345,459-463): the trunk
    runs under no_grad, its parameters still require grad but never receive one"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ssl4gie_amd.parallel import DataParallel
    torch.manual_seed(1234)
    m = _Toy()

    def probe_forward(x):
        m.arena()
        m._s.new_pass()
        with torch.no_grad():
            t = m._lin(x, m.stem) + m.pos * 2.0
            for blk in m.blocks:
                t = torch.tanh(m._lin(t, blk))
        return (m._lin(t, m.head.fc) ** 2).mean()

    m.forward = probe_forward
    model = DataParallel(m, device_ids=[0], find_unused_parameters=True, bucket_bytes=4 * 2000)
    batches = _batches(world, False)
    # single-process gradient of the mean loss, plain torch
    torch.manual_seed(1234)
    r = _Toy()
    tot = 0
    for b in batches:
        with torch.no_grad():
            t = b[0] @ r.stem.weight.t() + r.stem.bias + r.pos * 2.0
            for blk in r.blocks:
                t = torch.tanh(t @ blk.weight.t() + blk.bias)
        tot = tot + ((t @ r.head.fc.weight.t() + r.head.fc.bias) ** 2).mean() / len(batches)
    tot.backward()
    res = []
    for step in range(3):
        for p_ in m.parameters():
            p_.grad = None
        before = model.n_collectives
        loss = model(batches[rank][0])
        loss.backward()
        model.finish()
        ok = True
        for (k, p_), (_, e) in zip(m.named_parameters(), r.named_parameters()):
            if k.startswith("head."):
                ok &= p_.grad is not None and torch.allclose(p_.grad, e.grad, rtol=1e-5, atol=1e-7)
            else:
                ok &= p_.grad is None or float(p_.grad.abs().max()) == 0.0
        res.append((bool(ok), model.n_collectives - before, model.n_late))
    q.put((rank, res))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_frozen_trunk_under_no_grad_world2():
    """`frozen=True` wrappers under DataParallel(find_unused_parameters=True): every trunk parameter is unused in the
    learnt plan, the head's gradients are the rank mean, and every rank issues the same collectives per step"""
    out = sorted(_run(False, 0, 0, target=_probe_worker))
    assert out[0][1] == out[1][1] or [s[:2] for s in out[0][1]] == [s[:2] for s in out[1][1]]
    for rank, steps in out:
        for ok, ncoll, late in steps:
            assert ok and late == 0
        assert steps[1][1] == steps[2][1] and steps[1][1] <= 3, steps   # the head's slice + the unused tails


def _loop_worker(rank, world, port, q):
    """gradient accumulation under no_sync(), a parameter that turns up after it was learnt as unused,
    a backward pass that raises half-way, a module.-prefixed checkpoint round trip"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ssl4gie_amd import checkpoints
    from ssl4gie_amd.parallel import DataParallel
    torch.manual_seed(7)
    m = _Toy()
    model = DataParallel(m, bucket_bytes=4 * 2000)
    a = m.arena()
    batches = _batches(world, True)
    ok = True
    # --- accumulation: two micro-batches, the first without communication
    for p in m.parameters():
        p.grad = None
    x1, x2 = batches[rank]
    n0 = model.n_collectives
    with model.no_sync():
        model(x1).backward()
    ok &= model.n_collectives == n0
    model(x2).backward()
    exp = None
    for b in batches:
        g = _single_process_grads(7, [[b[0]]])
        g2 = _single_process_grads(7, [[b[1]]])
        tot = {k: (None if g[k] is None else (g[k] + g2[k]) / world) for k in g}
        exp = tot if exp is None else {k: (None if tot[k] is None else exp[k] + tot[k]) for k in tot}
    for k, p in m.named_parameters():
        if exp[k] is not None:
            ok &= torch.allclose(p.grad, exp[k], rtol=1e-5, atol=1e-7)
    # --- a parameter learnt as unused receives a gradient after all (the graph changed)
    for p in m.parameters():
        p.grad = None
    model(x1).backward()          # a plain pass: `unused` is known to be unused by now
    late0 = model.n_late
    for p in m.parameters():
        p.grad = None
    loss = model(x1) + (m.unused.weight ** 2).sum() * (rank + 1.0)
    loss.backward()
    ok &= model.n_late == late0 + 1
    ok &= torch.allclose(m.unused.weight.grad, 2 * m.unused.weight.detach() * 1.5, rtol=1e-6)
    ok &= m.unused.weight.grad.data_ptr() == a.grad_view(m.unused.weight).data_ptr()
    for p in m.parameters():
        p.grad = None
    n_before = model.n_collectives
    (model(x1) + (m.unused.weight ** 2).sum()).backward()   # still carried by its tail slice, every pass
    ok &= model.n_late == late0 + 2 and model.n_collectives > n_before
    ok &= torch.allclose(m.unused.weight.grad, 2 * m.unused.weight.detach(), rtol=1e-6)
    model.relearn()                                          # the schedule follows the new graph on request
    for p in m.parameters():
        p.grad = None
    (model(x1) + (m.unused.weight ** 2).sum()).backward()   # learning pass
    for p in m.parameters():
        p.grad = None
    (model(x1) + (m.unused.weight ** 2).sum()).backward()   # `unused` is part of the overlapped slices now
    ok &= model.n_late == late0 + 2
    ok &= torch.allclose(m.unused.weight.grad, 2 * m.unused.weight.detach(), rtol=1e-6)
    # --- a backward that raises leaves no stale pass behind
    class Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.clone()
        @staticmethod
        def backward(ctx, g):
            raise ValueError("boom")
    for p in m.parameters():
        p.grad = None
    try:
        Boom.apply(model(x1)).backward()
        ok = False
    except ValueError:
        pass
    for p in m.parameters():
        p.grad = None
    model.relearn()
    passes = model.n_passes
    model(x1).backward()
    ok &= model.n_passes == passes + 1
    # --- module.-prefixed file (main_moco.py:313) -> plain names (convert_to_deit.py:24-32)
    import io
    buf = io.BytesIO()
    torch.save({"state_dict": model.state_dict(), "epoch": 3}, buf)
    buf.seek(0)
    sd = torch.load(buf)["state_dict"]
    plain = checkpoints.ddp_unwrap(sd)
    torch.manual_seed(99)
    fresh = _Toy()
    fresh.load_state_dict(plain, strict=True)
    ok &= all(torch.equal(x, y) for x, y in zip(fresh.parameters(), m.parameters()))
    wrapped = DataParallel(_Toy(), broadcast_parameters=False)
    wrapped.load_state_dict(sd, strict=True)        # main_moco.py:246 loads INTO the wrapper
    ok &= all(torch.equal(x, y) for x, y in zip(wrapped.module.parameters(), m.parameters()))
    model.eval()
    ok &= not m.training
    model.train()
    ok &= m.training
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def _ragged_graph_worker(rank, world, port, q):
    """A rank whose graph differs from the others' in one pass (advisor finding, round 3): rank 1 skips one
    block in pass 2.  The slices are a plan (a function of the layout and the agreed unused set), so both
    ranks issue the same collectives of the same sizes in the same order — the pass completes, rank 1's
    skipped block contributes zeros, and the next (uniform) pass averages as before."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ssl4gie_amd.parallel import DataParallel
    torch.manual_seed(1234)
    m = _Toy()
    model = DataParallel(m, bucket_bytes=4 * 2000)     # several slices per pass
    a = m.arena()
    batches = _batches(world, False)
    x = batches[rank][0]
    ok = True

    def fwd(skip):
        m.arena(); m._s.new_pass()
        t = m._lin(x, m.stem) + m.pos * 2.0
        for i, blk in enumerate(m.blocks):
            if i == skip:
                continue
            t = torch.tanh(m._lin(t, blk))
        return (m._lin(t, m.head.fc) ** 2).mean()

    def plain_grads(xr, skip):
        torch.manual_seed(1234)
        r = _Toy()
        t = xr @ r.stem.weight.t() + r.stem.bias + r.pos * 2.0
        for i, blk in enumerate(r.blocks):
            if i == skip:
                continue
            t = torch.tanh(t @ blk.weight.t() + blk.bias)
        ((t @ r.head.fc.weight.t() + r.head.fc.bias) ** 2).mean().backward()
        return {k: p.grad for k, p in r.named_parameters()}

    sizes = []
    for step, skips in enumerate(((None, None), (None, 3), (None, None))):   # (rank 0, rank 1) per pass
        for p in m.parameters():
            p.grad = None
        a.grad.zero_()
        before = model.n_collectives
        model.forward  # (the wrapper's forward is bypassed on purpose: a hand-built graph per rank)
        fwd(skips[rank]).backward()
        sizes.append(model.n_collectives - before)
        g = [plain_grads(batches[r][0], skips[r]) for r in range(world)]
        for k, p in m.named_parameters():
            parts = [gi[k] for gi in g if gi[k] is not None]
            if not parts:
                continue
            exp = sum(parts) / world
            got = a.grad_view(p)
            ok &= torch.allclose(got, exp, rtol=1e-5, atol=1e-7)
    ok &= sizes[1] == sizes[2]       # same number of collectives whatever the rank's own graph did
    q.put((rank, bool(ok), sizes))
    dist.destroy_process_group()


def test_rank_local_graph_change_keeps_collectives_uniform_world2():
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ragged_graph_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(o[1] for o in out), out
    assert out[0][2] == out[1][2], "both ranks issued the same number of collectives in every pass"


def _nooverlap_worker(rank, world, port, q):
    """overlap=False (everything leaves at the end of backward, still inside it) and a bucket smaller than
    any parameter (one slice per parameter run): same averages"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ssl4gie_amd.parallel import DataParallel
    batches = _batches(world, False)
    expect = _single_process_grads(1234, batches)
    ok = True
    for kw in (dict(overlap=False), dict(bucket_bytes=4)):
        torch.manual_seed(1234)
        m = _Toy()
        model = DataParallel(m, **kw)
        for step in range(2):
            for p in m.parameters():
                p.grad = None
            before = model.n_overlapped
            model(*batches[rank]).backward()
            if "overlap" in kw:
                ok &= model.n_overlapped == before       # nothing leaves before the end of backward
            for k, p in m.named_parameters():
                if expect[k] is not None:
                    ok &= torch.allclose(p.grad, expect[k], rtol=1e-5, atol=1e-7)
        ok &= model.n_passes == 2
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_no_overlap_and_tiny_buckets_world2():
    res = _run(None, None, None, target=_nooverlap_worker)
    assert all(ok for _, ok in res), res


@pytest.mark.timeout(120)
def test_reference_loop_features_world2():
    res = _run(None, None, None, target=_loop_worker)
    assert all(ok for _, ok in res), res


def _infonce_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from types import SimpleNamespace
    import torch.nn.functional as F
    from ssl4gie_amd.Models.moco_v3.moco.builder import MoCo
    T, n = 0.2, 6
    g = torch.Generator().manual_seed(77)
    q_all = torch.randn(world * n, 16, generator=g, dtype=torch.float64)
    k_all = q_all + 0.5 * torch.randn(world * n, 16, generator=g, dtype=torch.float64)
    q = q_all[rank * n:(rank + 1) * n].clone().requires_grad_(True)
    k = k_all[rank * n:(rank + 1) * n].clone()
    loss = MoCo.contrastive_loss(SimpleNamespace(T=T), q, k)   # keys gathered, labels arange(n) + n * rank
    loss.backward()
    # single process, whole batch: every query against every key, label = its own index
    qa = q_all.clone().requires_grad_(True)
    logits = F.normalize(qa, dim=1) @ F.normalize(k_all, dim=1).t() / T
    ref = F.cross_entropy(logits, torch.arange(world * n)) * (2 * T)
    ref.backward()
    mean = loss.detach().clone()
    dist.all_reduce(mean)
    mean /= world                       # what DDP's gradient averaging and the logged loss see
    ok = abs(float(mean) - float(ref)) < 1e-12 * abs(float(ref))
    # DDP averages gradients over ranks; a query only lives on its own rank, so rank r's gradient / world
    # must be the whole-batch gradient's rows of that rank
    ok &= torch.allclose(q.grad / world, qa.grad[rank * n:(rank + 1) * n], rtol=1e-10, atol=1e-14)
    # wrong label offset (the bug the `+ n * rank` guards against) would give a different value on rank 1
    wrong = F.cross_entropy(F.normalize(q.detach(), dim=1) @ F.normalize(k_all, dim=1).t() / T, torch.arange(n)) * (2 * T)
    ok &= (rank == 0) == (abs(float(wrong) - float(loss)) < 1e-12)
    return bool(ok)


def _infonce_entry(rank, world, port, q):
    ok = _infonce_worker(rank, world, port, q)
    q.put((rank, ok))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_infonce_two_ranks_equals_single_process_on_the_concatenated_batch():
    """reference builder.py:63-73: keys all-gathered without gradient, labels arange(N) + N * rank —
    the mean over ranks of the per-rank loss is the whole-batch InfoNCE loss, gradients likewise"""
    res = _run(None, None, None, target=_infonce_entry)
    assert all(ok for _, ok in res), res


def _syncbn_worker(rank, world, port, q):
    import os
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ssl4gie_amd.resnet_engine import combine_batch_stats
    g = torch.Generator().manual_seed(5)
    full = torch.randn(7 + 13, 24, generator=g) * 3 + 2  # ranks hold 7 and 13 rows
    mine = full[:7] if rank == 0 else full[7:]
    mean, var, total = combine_batch_stats(mine.mean(0), mine.var(0, unbiased=False), mine.shape[0])
    ok = (torch.allclose(mean, full.mean(0), atol=1e-5) and torch.allclose(var, full.var(0, unbiased=False), atol=1e-4)
          and float(total) == 20.0)
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_syncbn_stat_combine_world2():
    """exchange step of SyncBatchNorm (uneven per-rank row counts) == statistics of the pooled rows"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29611
    procs = [ctx.Process(target=_syncbn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok in res), res


def _bt_worker(rank, world, port, q):
    import os
    import torch
    import torch.distributed as dist
    import torch.nn.functional as F
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ssl4gie_amd.Models.barlow_twins import cross_corr_loss_terms, exchange_cross_corr
    from oracle import bt_ref
    g = torch.Generator().manual_seed(9)
    z1 = torch.randn(24, 16, generator=g, dtype=torch.float64)
    z2 = z1 + 0.3 * torch.randn(24, 16, generator=g, dtype=torch.float64)
    ref = bt_ref.barlow_loss(z1, z2, 0.0051)
    # SyncBN semantics: normalise with the statistics of the pooled batch, then shard the rows
    zn1 = F.batch_norm(z1, None, None, None, None, True)
    zn2 = F.batch_norm(z2, None, None, None, None, True)
    rows = slice(0, 10) if rank == 0 else slice(10, 24)  # uneven shards
    c = zn1[rows].t() @ zn2[rows] / 24
    exchange_cross_corr(c)
    loss, dc = cross_corr_loss_terms(c, 0.0051)
    full = zn1.t() @ zn2 / 24
    ok = torch.allclose(c, full, atol=1e-12) and abs(float(loss) - float(ref)) < 1e-9 * abs(float(ref))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_barlow_twins_cross_corr_exchange_world2():
    """config 5's collective: per-rank correlation blocks summed over ranks == the pooled-batch
    correlation, and the loss on it == the oracle's single-process loss"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bt_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok in res), res


# ------------------------------------------------------------------ SSL4GIE_ALLREDUCE=auto: the transport probe
def test_choose_transport_is_a_pure_rank_uniform_verdict():
    from ssl4gie_amd.parallel import choose_transport
    ok = lambda ref, d, c=(1, 2.0): {"direct_ok": True, "ref_ms": ref, "direct_ms": d, "checksum": c}
    assert choose_transport([ok(2.0, 1.0), ok(2.1, 1.2)])[0] == "direct"
    assert choose_transport([ok(2.0, 1.0), ok(2.1, 2.5)])[0] == "rccl"            # max over ranks decides
    assert choose_transport([ok(2.0, 1.0), ok(2.0, 1.0, (1, 2.5))])[0] == "rccl"  # replicas would drift apart
    bad = {"direct_ok": False, "ref_ms": 2.0, "direct_ms": float("inf"), "checksum": None, "error": "run: timeout"}
    chosen, reason = choose_transport([ok(2.0, 1.0), bad])
    assert chosen == "rccl" and "[1]" in reason and "timeout" in reason


class _FakeDirect:
    """stands in for DirectAllReduce on gloo: a correct all-reduce(mean) with optional failure modes"""
    closed = 0

    def __init__(self, pg, fail_init=False, fail_run=False):
        if fail_init:
            raise RuntimeError("no peer access")
        self.pg, self.fail_run = pg, fail_run

    def set_timeout(self, s):
        self.timeout = s

    def all_reduce_(self, t, scale, stream=None):
        if self.fail_run:
            raise RuntimeError("rank 0 did not arrive in time")
        dist.all_reduce(t, group=self.pg)
        t.mul_(scale)

    def raise_if_failed(self):
        pass

    def close(self):
        _FakeDirect.closed += 1


def _transport_probe_worker(rank, world, port, q, case):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ssl4gie_amd.parallel import probe_transports
    kw = {}
    if case == "fault":
        kw["fault_rank"] = 1
    # "run": a peer never signals — on the real transport every rank's kernel then times out into the sticky error
    make = lambda: _FakeDirect(None, fail_init=(case == "init" and rank == 0), fail_run=(case == "run"))
    h, rep = probe_transports(None, "cpu", make, n_elems=1 << 16, iters=2, **kw)
    q.put((rank, h is not None, rep, _FakeDirect.closed))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("case", ["clean", "fault", "init", "run"])
def test_transport_probe_agrees_and_falls_back_in_process_world2(case):
    """the probe behind SSL4GIE_ALLREDUCE=auto with a stand-in transport: both ranks reach the same verdict and return;
    a corrupted result on one rank (the forced-mismatch hook), a handle that cannot be created on one rank and a
    peer that never arrives (the sticky-error path) all end in "rccl" on BOTH ranks, in this process, with the
    handle closed — never in a hang or a restart"""
    out = sorted(_run(False, 0, 0, target=_transport_probe_worker, extra=(case,)))
    (r0, has0, rep0, closed0), (r1, has1, rep1, closed1) = out
    assert rep0 == rep1 and has0 == has1 == (rep0["chosen"] == "direct")
    assert rep0["mib"] == 0.2 and rep0["rccl_ms"] is not None
    if case == "clean":
        assert rep0["chosen"] in ("direct", "rccl") and ("faster" in rep0["reason"])
    else:
        assert rep0["chosen"] == "rccl"
        assert {"fault": "rank(s) [1]", "init": "init: no peer access", "run": "did not arrive in time"}[case] in rep0["reason"]
        assert closed0 + closed1 >= 1   # whoever held a handle released it
