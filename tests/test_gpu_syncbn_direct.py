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


def _fused_worker(rank, world, port, q):
    """The single-process fusions carried through the SyncBatchNorm exchange (round 6): for each fused form, the 2-rank
    result against torch's BatchNorm on the POOLED batch (fp64) and against the un-fused SyncBatchNorm path
    (resnet_engine._SYNC_FUSED = False) of the same run."""
    import torch.distributed as dist
    import torch.nn.functional as F
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    out = {}
    try:
        from ssl4gie_amd import ops, resnet_engine
        from ssl4gie_amd.engine import GradSink
        from ssl4gie_amd.resnet_engine import BatchNormFn, BnReluMaxPoolFn, MaxPoolFn
        BF = torch.bfloat16
        rel = lambda a, b: float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max())

        def partials(x2):   # what the producing GEMM's epilogue emits: per-128-row sums / sums of squares
            xf = x2.float()
            xp = torch.cat([xf, xf.new_zeros((-xf.shape[0]) % 128, xf.shape[1])]).view(-1, 128, xf.shape[1])
            return torch.stack([xp.sum(1), (xp * xp).sum(1)], 1).contiguous()

        # ---- (i) bn3: BatchNorm + residual + ReLU with the ReLU bit map, uneven shards
        C, rows = 256, (384, 640)
        g = torch.Generator().manual_seed(11)
        full = (torch.randn(sum(rows), C, generator=g) * 1.5 + 0.5).to(BF)
        res_full = torch.randn(sum(rows), C, generator=g).to(BF)
        dy_full = torch.randn(sum(rows), C, generator=g).to(BF)
        gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.3 * torch.randn(C, generator=g)
        lo = sum(rows[:rank])
        sl = slice(lo, lo + rows[rank])
        vals = {}
        for fused in (True, False):
            resnet_engine._SYNC_FUSED = fused
            bn = torch.nn.SyncBatchNorm(C).cuda()
            with torch.no_grad():
                bn.weight.copy_(gamma)
                bn.bias.copy_(beta)
            x = full[sl].cuda().requires_grad_(True)
            r = res_full[sl].cuda().requires_grad_(True)
            y = BatchNormFn.apply(x, bn.weight, bn.bias, r, bn, True, GradSink(None), partials(x.detach()))
            y.backward(dy_full[sl].cuda())
            torch.cuda.synchronize()
            vals[fused] = (y.detach(), x.grad, r.grad, bn.weight.grad.clone(), bn.bias.grad.clone(),
                           bn.running_mean.clone(), bn.running_var.clone())
        out["bits_equal_unfused"] = all(torch.equal(a, b) for a, b in zip(vals[True], vals[False]))
        xr, rr = full.double().requires_grad_(True), res_full.double().requires_grad_(True)
        gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
        yr = F.relu(F.batch_norm(xr, torch.zeros(C, dtype=torch.float64), torch.ones(C, dtype=torch.float64), gr, br,
                                 True, 0.1, 1e-5) + rr)
        yr.backward(dy_full.double())
        out["bits_y"], out["bits_dx"], out["bits_dres"] = rel(vals[True][0], yr.detach()[sl]), rel(vals[True][1], xr.grad[sl]), \
            rel(vals[True][2], rr.grad[sl])
        gw = vals[True][3].cpu().clone(); dist.all_reduce(gw)
        out["bits_dgamma"] = rel(gw, gr.grad)
        # ---- (ii) stem: bn1 -> relu -> maxpool in one pass
        B, H, W, C2 = (2, 3)[rank], 18, 22, 64
        g = torch.Generator().manual_seed(21 + rank)
        xm = (torch.randn(B, H, W, C2, generator=g) * 1.4 + 0.2).to(BF).cuda()
        vals = {}
        for fused in (True, False):
            resnet_engine._SYNC_FUSED = fused
            bn = torch.nn.SyncBatchNorm(C2).cuda()
            with torch.no_grad():
                bn.weight.copy_(1 + 0.2 * torch.randn(C2, generator=torch.Generator().manual_seed(5)))
                bn.bias.copy_(0.3 * torch.randn(C2, generator=torch.Generator().manual_seed(6)))
            xi = xm.clone().requires_grad_(True)
            st = partials(xi.detach().view(-1, C2))
            if fused:
                assert resnet_engine.bn_relu_maxpool_ok(xi, bn, st)
                y = BnReluMaxPoolFn.apply(xi, bn.weight, bn.bias, bn, GradSink(None), st)
            else:
                assert not resnet_engine.bn_relu_maxpool_ok(xi, bn, st)
                y = MaxPoolFn.apply(BatchNormFn.apply(xi, bn.weight, bn.bias, None, bn, True, GradSink(None), st))
            y.backward(torch.randn(y.shape, generator=torch.Generator().manual_seed(7 + rank)).to(BF).cuda())
            torch.cuda.synchronize()
            vals[fused] = (y.detach(), xi.grad, bn.weight.grad.clone(), bn.bias.grad.clone(), bn.running_mean.clone(),
                           bn.running_var.clone())
        out["pool_equal_unfused"] = all(torch.equal(a, b) for a, b in zip(vals[True], vals[False]))
        out["pool_maxdiff"] = [float((a.float() - b.float()).abs().max()) for a, b in zip(vals[True], vals[False])]
        # ---- (iii) the no-grad trunk (MoCo's momentum encoder): statistics-only product -> exchange -> product with
        # the normalisation in its epilogue, for the widening 1x1 convolutions, and the stem pool pass, end to end
        from ssl4gie_amd.Models.resnet import ResNet50
        torch.manual_seed(5)
        net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(ResNet50()).cuda()
        net.set_precision("bf16")
        sd0 = {k: v.clone() for k, v in net.state_dict().items()}
        # bs 16 of 64 x 64: layer1 / layer2 hold >= 128 rows per column statistics tile (ops.colstats_ok)
        ximg = torch.randn(16, 3, 64, 64, generator=torch.Generator().manual_seed(31 + rank)).cuda()
        vals = {}
        for fused in (True, False):
            resnet_engine._SYNC_FUSED = fused
            net.load_state_dict(sd0)
            net.train()
            with torch.no_grad():
                maps = net.forward_maps(ximg, all_stages=True)
            torch.cuda.synchronize()
            vals[fused] = [m_.float() for m_ in maps] + [net.layer2[0].downsample[1].running_var.clone(),
                                                         net.layer1[0].bn3.running_mean.clone()]
        l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
        out["nograd_rel"] = [l2(a, b) for a, b in zip(vals[True], vals[False])]
        resnet_engine._SYNC_FUSED = True
        out["dist_calls"] = resnet_engine.SYNC_BN_COLLECTIVES[0]
    except Exception:  # noqa: BLE001
        import traceback
        out["error"] = traceback.format_exc()
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_syncbn_fused_forms_two_processes_one_device():
    """BASELINE config 3 is 8-GPU MoCo with SyncBatchNorm (Models/moco_v3/main_moco.py:196): the bit-map backward of
    bn3 and the stem's bn -> relu -> maxpool pass must survive the exchange — equal to the un-fused SyncBatchNorm path
    bit for bit, and right against torch's BatchNorm on the pooled batch"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fused_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for r in (0, 1):
        o = res[r]
        assert "error" not in o, o["error"]
        assert o["bits_equal_unfused"], "bn3 bit-map form != separate passes under SyncBatchNorm"
        assert o["pool_equal_unfused"], ("stem pool form != separate passes under SyncBatchNorm", o["pool_maxdiff"])
        assert o["bits_y"] < 1e-2 and o["bits_dx"] < 2e-2 and o["bits_dres"] < 1e-2 and o["bits_dgamma"] < 1e-2, o
        # the affine epilogue normalises the fp32 accumulators (one bf16 rounding less than the separate pass): the
        # first stage and the first blocks' running statistics are tight; 16 random-init bottlenecks at 128-2048 rows
        # per statistic are ill-conditioned (tests/test_gpu_resnet.py allows 0.3 on one rank for the same comparison)
        nr = o["nograd_rel"]
        # (layer3 / layer4 maps are 4 x 4 / 2 x 2 here: 512 / 128 rows per statistic over both ranks — measured 0.15 / 0.32)
        assert nr[0] < 3e-2 and nr[1] < 0.1 and nr[2] < 0.3 and nr[3] < 0.5 and max(nr[4:]) < 2e-3, nr
