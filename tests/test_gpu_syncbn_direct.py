"""GPU: SyncBatchNorm's exchange (reference `convert_sync_batchnorm`, Models/moco_v3/main_moco.py:196,
Depth_estimation/train_depth.py:225) carried by the library's peer-to-peer all-gather
(`SSL4GIE_SYNCBN=direct`: ssl4gie_allgather_direct_enqueue + ssl4gie_bn_combine_stats, no RCCL launch) —
two processes on the one device, UNEVEN row counts per rank: output, input gradient, parameter gradients
and running statistics equal torch's BatchNorm on the concatenated batch, and equal the torch.distributed
(gloo here) path bit for bit in what the ranks agree on."""
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


def _worker(rank, world, port, q, mode):
    import torch.distributed as dist
    import torch.nn.functional as F
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SSL4GIE_SYNCBN=mode)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    out = {}
    try:
        from ssl4gie_amd import resnet_engine
        from ssl4gie_amd.engine import GradSink
        from ssl4gie_amd.resnet_engine import BatchNormFn
        C, rows = 96, (40, 88)      # uneven shards
        g = torch.Generator().manual_seed(3)
        full = torch.randn(sum(rows), C, generator=g) * 2 + 1
        dy_full = torch.randn(sum(rows), C, generator=g)
        gamma = 1 + 0.2 * torch.randn(C, generator=g)
        beta = 0.3 * torch.randn(C, generator=g)
        lo = sum(rows[:rank])
        x = full[lo:lo + rows[rank]].cuda().requires_grad_(True)
        bn = torch.nn.SyncBatchNorm(C).cuda()
        with torch.no_grad():
            bn.weight.copy_(gamma)
            bn.bias.copy_(beta)
        for step in range(2):   # twice: parity reuse of the exchange regions, running statistics move twice
            x.grad = None
            bn.weight.grad = bn.bias.grad = None
            y = BatchNormFn.apply(x, bn.weight, bn.bias, None, bn, True, GradSink(None))
            y.backward(dy_full[lo:lo + rows[rank]].cuda())
        torch.cuda.synchronize()
        # single-process reference on the pooled batch (fp64)
        xr = full.double().requires_grad_(True)
        gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
        rm, rv = torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64)
        for step in range(2):
            yr = F.relu(F.batch_norm(xr, rm, rv, gr, br, True, 0.1, bn.eps))
        yr.backward(dy_full.double())
        rel = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max())
        out["y"] = rel(y.detach(), yr.detach()[lo:lo + rows[rank]])
        out["dx"] = rel(x.grad, xr.grad[lo:lo + rows[rank]])
        out["rm"] = rel(bn.running_mean, rm)
        out["rv"] = rel(bn.running_var, rv)
        # parameter gradients are LOCAL sums (DDP averages them): their sum over ranks is the pooled gradient
        gw = bn.weight.grad.detach().cpu().clone()
        gb = bn.bias.grad.detach().cpu().clone()
        dist.all_reduce(gw)
        dist.all_reduce(gb)
        out["dgamma"] = rel(gw, gr.grad)
        out["dbeta"] = rel(gb, br.grad)
        out["direct_calls"] = resnet_engine.SYNC_BN_DIRECT[0]
        out["dist_calls"] = resnet_engine.SYNC_BN_COLLECTIVES[0]
        out["vals"] = (y.detach().cpu().numpy(), x.grad.cpu().numpy(), bn.running_var.cpu().numpy())
    except Exception:  # noqa: BLE001
        import traceback
        out["error"] = traceback.format_exc()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_syncbn_direct_exchange_two_processes_one_device():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    res = {}
    for mode in ("dist", "direct"):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, q, mode)) for r in range(2)]
        for p in procs:
            p.start()
        res[mode] = dict(q.get(timeout=240) for _ in procs)
        for p in procs:
            p.join(timeout=60)
    for mode in ("dist", "direct"):
        for r in (0, 1):
            o = res[mode][r]
            assert "error" not in o, o["error"]
            for k in ("y", "dx", "rm", "rv", "dgamma", "dbeta"):
                assert o[k] < 2e-5, (mode, r, k, o[k])
    assert res["direct"][0]["direct_calls"] == 4 and res["direct"][0]["dist_calls"] == 0   # 2 steps x (fwd + bwd)
    assert res["dist"][0]["direct_calls"] == 0 and res["dist"][0]["dist_calls"] == 4
    for r in (0, 1):   # the two transports agree to fp32 rounding of the combine
        for a, b in zip(res["dist"][r]["vals"], res["direct"][r]["vals"]):
            assert np.allclose(a, b, rtol=1e-5, atol=1e-6)
