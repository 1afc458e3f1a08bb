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


class _Toy(nn.Module):
    """Stands in for an EngineModule: parameters in registration order, arena, _grad_hook."""

    def __init__(self):
        super().__init__()
        self.stem = nn.Linear(8, 32)
        self.blocks = nn.ModuleList([nn.Linear(32, 32) for _ in range(6)])
        self.head = nn.Linear(32, 4)
        self._grad_hook = None
        self._a = None

    def arena(self):
        from ssl4gie_amd.engine import ParamArena
        if self._a is None:
            self._a = ParamArena(list(self.parameters()))
        return self._a


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ssl4gie_amd.parallel import DataParallel
    torch.manual_seed(1234 + rank)  # different initial weights per rank: broadcast must fix it
    m = _Toy()
    ddp = DataParallel(m, bucket_bytes=4 * 2000, overlap=True)
    a = m.arena()
    w0 = a.data.clone()
    gather = [torch.empty_like(w0) for _ in range(world)]
    dist.all_gather(gather, w0)
    same_weights = all(torch.equal(gather[0], g) for g in gather)
    # emulate a backward that fills the arena from the end (head first, stem last)
    gen = torch.Generator().manual_seed(77 + rank)
    local = torch.randn(a.grad.numel(), generator=gen)
    a.grad.copy_(local)
    for blk in reversed(m.blocks):
        m._grad_hook(blk)
    fired_during_backward = ddp.n_collectives
    ddp.finish()
    expect = sum(torch.randn(a.grad.numel(), generator=torch.Generator().manual_seed(77 + r))
                 for r in range(world)) / world
    ok = torch.allclose(a.grad, expect, rtol=0, atol=1e-6)
    loss_mean = ddp.all_reduce_mean(torch.tensor([float(rank + 1)]))
    q.put((rank, same_weights, bool(ok), fired_during_backward, ddp.n_collectives,
           float(loss_mean)))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_bucketed_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=100) for _ in procs]
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, same_w, ok, during, total, lm in res:
        assert same_w, "rank-0 parameter broadcast"
        assert ok, "gradient average"
        assert during >= 2, "buckets must be launched while backward is still running"
        assert total == during + 1
        assert abs(lm - 1.5) < 1e-6


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
