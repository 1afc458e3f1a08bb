#!/usr/bin/env python3
"""Headline benchmark: MAE ViT-B pretraining step (BASELINE.json configs[1]) on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = forward + backward of MaskedAutoencoderViT (ViT-B encoder on 25 % of the tokens, 8x512-d
decoder on all 197, masked-MSE loss with norm_pix_loss) on a fixed synthetic batch of 256 images
per GPU (N(0,1) pixels, resident in HBM), gradient averaging across ranks (bucketed RCCL all-reduce
overlapped with backward, ssl4gie_amd.parallel), AdamW(b=(0.9,0.95), wd 0.05) step.  bf16 MFMA
operands, fp32 accumulate / residual / statistics.  Prints ONE JSON line on rank 0.

Extra objects on the line:
  roofline      dominant kernel (bf16 NT GEMM, v_mfma_f32_16x16x32_bf16): algorithmic FLOPs (2MNK
                summed over its launches) / summed launch durations, measured with HIP events on
                the launch stream by the library's launch profiler over `--prof-steps` further
                steps of the same workload (kept out of the timed region so that event records do
                not perturb `value`; the weight-gradient side stream is folded back onto the main
                stream for those steps so that every launch is timed alone on the chip).  peak = 2.5 PFLOP/s dense bf16 (MI355X_MICROARCH.md).
  cpu_baseline  the CPU oracle (oracle/mae_ref.py: fp32 torch restatement of the reference's
                MaskedAutoencoderViT step, validated against reference-generated fixtures) timed
                on this box's host cores at bs=8 (BASELINE.json configs[0]); rank 0, N=1 only.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = "images/sec/GPU (fwd+bwd) ViT-B 224² bs=256 at 1/2/4/8 MI355X; % MFMA peak"
GFLOP_PER_IMG = 58.69     # MAE ViT-B fwd+bwd, GEMM terms only (SURVEY §8d / BASELINE.md §2)
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def dp_verify(ddp, step, a, dev):
    """The N > 1 line proves itself (collective: every rank calls it, right after the timed region):
      backend_world      what the backend itself says: an all-reduce of ones on the device must sum to N
      replicas_in_sync   MIN and MAX over ranks of a checksum of the parameter arena (fp64 sum and sum of
                         squares) after the timed steps are equal — every rank applied the same averaged
                         gradients (bitwise: ring / one-hop all-reduce deliver identical sums to every rank)
      exposed_comm_ms    step time minus the same step under no_sync() (no gradient exchange), a few steps
                         each, barrier + synchronize around both: the communication NOT hidden behind backward
    A run whose replicas drifted apart is not a data-parallel measurement: main() exits non-zero on it
    (reference semantics being checked: DDP's gradient averaging, main_pretrain.py:175, train_depth.py:47-48)."""
    if ddp is None:
        return None
    import torch.distributed as dist
    cuda = dev.type == "cuda"

    def fence():
        dist.barrier()
        if cuda:
            torch.cuda.synchronize()

    ones = torch.ones(1, dtype=torch.float32, device=dev)
    dist.all_reduce(ones)
    if os.environ.get("SSL4GIE_BENCH_DESYNC_RANK") == str(dist.get_rank()):  # test hook: break one replica
        ddp.module.arena().data[:1].add_(1.0)
    w = ddp.module.arena().data.double()
    chk = torch.stack([w.sum(), (w * w).sum()])
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    in_sync = bool(torch.equal(lo, hi)) and bool(torch.isfinite(chk).all())
    k = max(1, min(a.steps, 5))
    fence()
    t0 = time.perf_counter()
    for _ in range(k):
        step()
    fence()
    t_sync = time.perf_counter() - t0
    with ddp.no_sync():
        t0 = time.perf_counter()
        for _ in range(k):
            step()
        fence()
        t_nosync = time.perf_counter() - t0
    tt = torch.tensor([t_sync, t_nosync], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return {"backend_world": int(round(float(ones.item()))), "backend": str(dist.get_backend()),
            "replicas_in_sync": in_sync,
            "param_checksum": [float(lo[0]), float(hi[0])],
            "exposed_comm_ms": round(1e3 * float(tt[0] - tt[1]) / k, 3),
            "step_ms_without_exchange": round(1e3 * float(tt[1]) / k, 3),
            "compute_cus": getattr(ddp, "compute_cus", None)}


def tree_stamp():
    """the commit libssl4gie_hip.so was built from (__graft_entry__.build() writes it; None when never stamped)"""
    try:
        with open(os.path.join(ROOT, "ssl4gie_amd", "_tree.txt")) as f:
            return f.read().strip() or None
    except OSError:
        return None


def dp_exit_if_broken(verify, world):
    """after rank 0 has printed its line: every rank leaves with a failure code when the run did not prove
    itself (the verdict is all-reduced, so all ranks agree; the launcher reports the non-zero exit)"""
    if verify is not None and (not verify["replicas_in_sync"] or verify["backend_world"] != world):
        sys.stderr.write(f"bench.py: data-parallel self-check FAILED: {verify}\n")
        sys.exit(3)


def dp_info(ddp, steps_run, verify=None):
    """data-parallel bookkeeping for the N > 1 lines: gradient collectives per step and how many of
    them left while backward was still running, parameters that had to be reduced on their own,
    what carried the slices (rccl / direct / gloo in rehearsals), SyncBatchNorm exchanges per step,
    and the self-check of dp_verify()"""
    if ddp is None:
        return {}
    from ssl4gie_amd import resnet_engine
    n = max(steps_run, 1)
    verify = dict(verify or {})
    snap = verify.pop("_snap", None) or (ddp.n_collectives, ddp.n_overlapped, ddp.n_late, ddp.n_passes,
                                         resnet_engine.SYNC_BN_COLLECTIVES[0], resnet_engine.SYNC_BN_DIRECT[0])
    return {"dp": {**verify,
                   "grad_collectives_per_step": round(snap[0] / n, 2),
                   "overlapped_with_backward_per_step": round(snap[1] / n, 2),
                   "late_params": snap[2], "transport": ddp.transport,
                   "transport_probe": getattr(ddp, "transport_probe", None),
                   "syncbn_probe": (list(__import__("ssl4gie_amd.parallel", fromlist=["x"]).SYNCBN_PROBE.values()) or [None])[0],
                   "passes_closed_in_backward": snap[3],
                   "syncbn_collectives_per_step": round(snap[4] / n, 1),
                   "syncbn_direct_exchanges_per_step": round(snap[5] / n, 1)}}


def self_launch(argv):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start the N ranks ourselves,
    BEFORE anything in this process touches the GPU (a process that has initialised HIP must never be
    exec-replaced, and the parent stays a plain supervisor), as children of
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`; rank 0's
    JSON line goes to our stdout, the exit code is the launcher's (non-zero if any rank failed)."""
    import socket
    import subprocess
    n = None
    for i, t in enumerate(argv):
        if t == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif t.startswith("--gpus="):
            n = int(t.split("=", 1)[1])
    if not n or n <= 1 or "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return None
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def param_groups(model, wd=0.05):
    """timm add_weight_decay semantics used by the reference driver (main_pretrain.py:179)."""
    decay, no_decay = [], []
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        (no_decay if (p.ndim <= 1 or n.endswith(".bias")) else decay).append(p)
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": wd}]


def host_cores():
    """threads for the CPU baseline: the physical cores this process may actually use — the smaller
    of the affinity mask, the distinct (package, core) pairs in /proc/cpuinfo (no SMT siblings) and
    the cgroup CPU quota (a GPU box hands one job a share of the host, e.g. 16 CPUs of 128: running
    128 threads on that share is what made round 1's baseline slower than 8 threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        phys, pkg = set(), "0"
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                pkg = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                phys.add((pkg, ln.split(":")[1].strip()))
        if phys:
            n = min(n, len(phys))
    except OSError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    env = os.environ.get("SSL4GIE_CPU_THREADS")
    return max(1, int(env)) if env else max(1, n)


def mae_nt_algorithmic_bytes(B, enc=(50, 768, 3072, 12), dec=(197, 512, 2048, 8)):
    """HBM bytes the NT GEMM launches of one MAE step must move if every operand is read once and
    every output written once (bf16 operands / activations, fp32 residual stream), averaged per
    launch: per block forward qkv, proj(+residual), fc1(2 outputs), fc2(+residual) and the four
    data-gradient products (dfc2 reads the saved GELU derivative)."""
    total, launches = 0, 0
    for n, D, F, depth in (enc, dec):
        T = B * n
        w = lambda a, b: 2 * a * b
        per = 0
        per += 2 * T * D + w(3 * D, D) + 2 * T * 3 * D                 # qkv
        per += 2 * T * D + w(D, D) + 4 * T * D + 4 * T * D             # proj + fp32 residual in/out
        per += 2 * T * D + w(F, D) + 2 * 2 * T * F                     # fc1 -> gelu', gelu
        per += 2 * T * F + w(D, F) + 4 * T * D + 4 * T * D             # fc2 + residual
        per += 2 * T * D + w(D, F) + 2 * T * F + 2 * T * F             # dfc2 (aux read, du write)
        per += 2 * T * F + w(F, D) + 2 * T * D                         # dfc1
        per += 2 * T * D + w(D, D) + 2 * T * D                         # dproj
        per += 2 * T * 3 * D + w(3 * D, D) + 2 * T * D                 # dqkv
        total += per * depth
        launches += 8 * depth
    T = B * 50
    total += 2 * T * 768 + 2 * 768 * 768 + 2 * T * 768                 # patch embed (kept patches)
    total += 2 * T * 768 + 2 * 512 * 768 + 2 * T * 512                 # decoder_embed
    total += 2 * T * 512 + 2 * 768 * 512 + 2 * T * 768                 # d decoder_embed
    Td = B * 197
    total += 2 * Td * 512 + 2 * 768 * 512 + 4 * Td * 768               # decoder_pred (fp32 out)
    total += 2 * Td * 768 + 2 * 768 * 512 + 2 * Td * 512               # d decoder_pred
    launches += 5
    return total, launches


def cpu_baseline(steps=10, warmup=3, b=8):  # 3 warm-up + 10 timed steps
    """CPU oracle step (fwd + bwd + AdamW) on the host cores; bounded sample (~10-30 s)."""
    from oracle import mae_ref, synth
    torch.set_num_threads(host_cores())
    cfg = mae_ref.MAEConfig(**{**mae_ref.VIT_B.__dict__, "norm_pix_loss": True})
    sd = synth.mae_state_dict(cfg, 0)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if "pos_embed" not in k}
    full = dict(sd)
    full.update(params)
    decay = [v for k, v in params.items() if v.ndim > 1 and not k.endswith(".bias")]
    nodecay = [v for k, v in params.items() if not (v.ndim > 1 and not k.endswith(".bias"))]
    opt = torch.optim.AdamW([{"params": nodecay, "weight_decay": 0.0},
                             {"params": decay, "weight_decay": 0.05}], lr=1.5e-4, betas=(0.9, 0.95))
    imgs = synth.synth_images(b, cfg, seed=0)
    noise = synth.synth_noise(b, 196, seed=0)
    t0 = None
    for it in range(warmup + steps):
        if it == warmup:
            t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss, _, _, _ = mae_ref.mae_forward(full, cfg, imgs, noise)
        loss.backward()
        opt.step()
    dt = time.perf_counter() - t0
    return {"value": round(b * steps / dt, 3), "unit": "images/sec", "cores": torch.get_num_threads(),
            "kind": "port", "s_per_step": round(dt / steps, 4),
            "sample": f"{steps} steps of MAE ViT-B fwd+bwd+AdamW at bs={b}, fp32, "
                      f"CPU oracle (oracle/mae_ref.py), {warmup} warm-up steps"}


PROF_KIND_NAMES = ["gemm_bf16_nt", "gemm_bf16_tn", "attn_fwd_bf16", "attn_bwd_bf16", "gemm_generic",
                   "batchnorm", "layernorm"]
PROF_HBM_KINDS = {"batchnorm", "layernorm"}   # these carry algorithmic BYTES in the profiler's work slot


def roofline_pass(step, a, workload):
    """`--prof-steps` more steps of the workload with the library's launch profiler on (HIP events on the launch
    stream around every launch of the heavy kernel kinds; outside the timed region, every stream folded back so
    that a launch has the chip to itself) -> the `roofline` object of the bench line for the kind with the largest
    summed duration: bound "mfma" (GEMM / attention kinds: algorithmic FLOPs / duration against 2.5 PFLOP/s) or
    "hbm" (BatchNorm / LayerNorm entry points: algorithmic bytes / duration against 8 TB/s)."""
    if a.prof_steps <= 0:
        return None
    from ssl4gie_amd import _lib
    from ssl4gie_amd import engine as _engine
    L = _lib.load()
    kinds = PROF_KIND_NAMES
    nk = _lib.PROF_KINDS
    L.ssl4gie_set_wgrad_stream(0)
    prev_side = _engine.set_wgrad_side(False)   # ... and the single layers' one (engine.wgrad_fork)
    _lib.check(L.ssl4gie_prof_begin(6000 * a.prof_steps), "prof_begin")
    for _ in range(a.prof_steps):
        step()
    ms = (ctypes.c_double * nk)()
    fl = (ctypes.c_double * nk)()
    nl = (ctypes.c_longlong * nk)()
    _lib.check(L.ssl4gie_prof_collect(ms, fl, nl), "prof_collect")
    L.ssl4gie_prof_end()
    L.ssl4gie_set_wgrad_stream(0 if os.environ.get("SSL4GIE_WGRAD_STREAM") == "0" else 1)
    _engine.set_wgrad_side(prev_side)
    per = {}
    for i, k in enumerate(kinds):
        if not nl[i]:
            continue
        rate = fl[i] / max(ms[i], 1e-9) / 1e9  # TFLOP/s, or TB/s for the byte kinds
        per[k] = {"launches_per_step": nl[i] // a.prof_steps, "ms_per_step": round(ms[i] / a.prof_steps, 3),
                  "avg_launch_us": round(1e3 * ms[i] / max(nl[i], 1), 2)}
        per[k]["GBs" if k in PROF_HBM_KINDS else "tflops"] = round(rate * (1e3 if k in PROF_HBM_KINDS else 1), 1)
    dom = max(range(nk), key=lambda i: ms[i])
    name = kinds[dom]
    rate = fl[dom] / max(ms[dom], 1e-9) / 1e9
    # HBM bytes per launch of that kernel kind from the committed rocprofv3 --pmc passes (FETCH_SIZE doubled per
    # the gfx950 note + WRITE_SIZE); measured for the MAE workload's kinds only, null otherwise
    traffic, traffic_from = None, None
    if workload == "mae":
        try:   # ONE committed file (tools/pmc_to_json.py), stamped with the tree its counters were taken on
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pj = json.load(f)
            traffic = pj["hbm_bytes_per_launch"].get(name)
            traffic_from = {"file": "profiles/pmc_traffic.json", "tree": pj.get("tree"), "source": pj.get("source")}
        except Exception:
            traffic = None
    if name in PROF_HBM_KINDS:
        return {"bound": "hbm", "kernel": name, "achieved": round(rate * 1e3, 1), "peak": PEAK_HBM_GBS,
                "unit": "GB/s", "frac": round(rate * 1e3 / PEAK_HBM_GBS, 4), "traffic": traffic, "traffic_from": traffic_from,
                "algorithmic_bytes_per_launch": round(fl[dom] / max(nl[dom], 1)),
                "avg_launch_us": per[name]["avg_launch_us"], "launches_per_step": per[name]["launches_per_step"],
                "kernels": per}
    # the same kernel against the HBM roof (these GEMMs sit near the ridge: K = 512 / 768 makes the epilogue
    # traffic first-order): bytes per launch / average launch duration / 8 TB/s
    hbm = None
    if name == "gemm_bf16_nt" and workload == "mae":
        alg_b, alg_n = mae_nt_algorithmic_bytes(a.batch)
        avg_s = 1e-3 * ms[dom] / max(nl[dom], 1)
        hbm = {"algorithmic_bytes_per_launch": round(alg_b / alg_n),
               "achieved_algorithmic_GBs": round(alg_b / alg_n / avg_s / 1e9, 1),
               "achieved_measured_GBs": round(traffic / avg_s / 1e9, 1) if traffic else None,
               "peak_GBs": PEAK_HBM_GBS,
               "frac_algorithmic": round(alg_b / alg_n / avg_s / 1e9 / PEAK_HBM_GBS, 4),
               "frac_measured": round(traffic / avg_s / 1e9 / PEAK_HBM_GBS, 4) if traffic else None}
    # the largest HBM-bound kind beside it (MoCo-R50: the BatchNorm passes are 1/3 of the kernel time)
    hb = [i for i, k in enumerate(kinds) if k in PROF_HBM_KINDS and nl[i]]
    hbm_kernel = None
    if hb:
        j = max(hb, key=lambda i: ms[i])
        r = fl[j] / max(ms[j], 1e-9) / 1e6  # GB/s
        hbm_kernel = {"bound": "hbm", "kernel": kinds[j], "achieved": round(r, 1), "peak": PEAK_HBM_GBS,
                      "unit": "GB/s", "frac": round(r / PEAK_HBM_GBS, 4), "ms_per_step": round(ms[j] / a.prof_steps, 3),
                      "algorithmic_bytes_per_step": round(fl[j] / a.prof_steps)}
    return {"bound": "mfma", "kernel": name, "achieved": round(rate, 1), "peak": PEAK_BF16_TFLOPS,
            "hbm_bound_kernel": hbm_kernel,
            "unit": "TFLOP/s", "frac": round(rate / PEAK_BF16_TFLOPS, 4), "hbm_roof": hbm, "traffic": traffic, "traffic_from": traffic_from,
            "traffic_unit": "HBM bytes/launch (profiles/pmc_traffic.json)",
            "flops_per_launch": round(fl[dom] / max(nl[dom], 1)), "avg_launch_us": per[name]["avg_launch_us"],
            "launches_per_step": per[name]["launches_per_step"], "kernels": per}


def timed_steps(step, a, world, dev, ddp=None):
    """the driver's timing contract: W untimed warm-up steps, then exactly K steps bracketed by a
    barrier + torch.cuda.synchronize() on both sides; returns (last loss, MAX over ranks of the
    elapsed seconds, the data-parallel self-check of dp_verify() — None for one rank — taken after the
    timed region)"""
    import torch.distributed as dist

    def fence():
        if world > 1:
            dist.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize()

    loss = None
    for _ in range(a.warmup):
        loss = step()
    # per-step durations beside the contract's one interval ("hipEvents around each step, report
    # median"): an event on the step's stream at every step boundary — recorded inside the timed region, never
    # waited for there
    cuda = dev.type == "cuda"
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)] if cuda else None
    fence()
    t0 = time.perf_counter()
    host = []
    for i in range(a.steps):
        if cuda:
            evs[i].record()
        h0 = time.perf_counter()
        loss = step()
        host.append(time.perf_counter() - h0)
    if cuda:
        evs[a.steps].record()
    fence()
    dt = time.perf_counter() - t0
    # host time inside step(): the first timed step starts on an idle GPU with empty queues, so its host time is the
    # pure enqueue cost of a step (no back-pressure from a full queue); well under ms_per_step = the GPU is the bound
    a.host_enqueue_ms = round(1e3 * host[0], 3) if host else None
    a.median_ms_per_step = None
    if cuda:
        per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(a.steps))
        n = len(per)
        a.median_ms_per_step = round(per[n // 2] if n % 2 else 0.5 * (per[n // 2 - 1] + per[n // 2]), 3)
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    snap = None
    if ddp is not None:  # the per-step figures of dp_info() describe the warm-up + timed steps only
        from ssl4gie_amd import resnet_engine
        snap = (ddp.n_collectives, ddp.n_overlapped, ddp.n_late, ddp.n_passes,
                resnet_engine.SYNC_BN_COLLECTIVES[0], resnet_engine.SYNC_BN_DIRECT[0])
    verify = dp_verify(ddp, step, a, dev)
    if verify is not None:
        verify["_snap"] = snap
    return loss, float(tt.item()), verify


def bench_depth(a):
    """configs[3]: ViT_from_MAE(dense="depth") + SSI loss + AdamW(1e-4) (train_depth.py:22-78,230,280)
    on synthetic img [B,3,224,224] N(0,1) and depth targets U(0,1) with 10 % zeros."""
    import torch.distributed as dist
    from ssl4gie_amd import _lib, parallel
    from ssl4gie_amd.Models import models
    from ssl4gie_amd.losses import ScaleAndShiftInvariantLoss
    _lib.load()
    rank, local, world = parallel.init_from_env()
    assert world == a.gpus
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    B = a.batch if a.batch != 256 else 128
    torch.manual_seed(0)
    model = models.ViT_from_MAE(None, False, 1, False, "depth", False, None, 768, 12, 12, "cls")
    model.to(dev).set_precision(a.precision)
    ddp = parallel.DataParallel(model) if world > 1 else None
    if a.optim == "arena":
        from ssl4gie_amd.optim import ArenaAdamW
        opt = ArenaAdamW(model, [p for p in model.parameters() if p.requires_grad], lr=1e-4)
    else:
        opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4, fused=True)
    loss_fn = ScaleAndShiftInvariantLoss(alpha=0.1)
    g = torch.Generator("cpu").manual_seed(rank)
    imgs = torch.randn(B, 3, 224, 224, generator=g).to(dev)
    tgt = torch.rand(B, 1, 224, 224, generator=g)
    tgt = torch.where(torch.rand(B, 1, 224, 224, generator=g) < 0.1, torch.zeros(()), tgt).to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = loss_fn((ddp or model)(imgs), tgt)
        loss.backward()   # under DataParallel the gradient exchange completes in here
        opt.step()
        return loss

    loss, dt, verify = timed_steps(step, a, world, dev, ddp)
    roof = roofline_pass(step, a, a.workload)
    if world > 1:
        dist.barrier()
    if rank == 0:
        ips = B * world * a.steps / dt
        print(json.dumps({**dp_info(ddp, a.steps + a.warmup, verify), 
            "metric": "images/sec (fwd+bwd+AdamW) ViT-B + DPT depth finetune 224x224 (BASELINE.json configs[3])",
            "value": round(ips, 1), "unit": "images/sec", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
            "median_ms_per_step": a.median_ms_per_step, "host_enqueue_ms": a.host_enqueue_ms, "roofline": roof, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
            "config": {"workload": "ViT_from_MAE(dense='depth') + DPT_decoder + SSI loss(alpha=0.1) + "
                                   "AdamW(1e-4), synthetic img + depth resident in HBM",
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}"},
            "model_mfma_frac": round(ips / world * 214.86 / 1e3 / PEAK_BF16_TFLOPS, 4),
            "final_loss": round(float(loss.detach()), 5)}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    dp_exit_if_broken(verify, world)


def bench_moco(a):
    """configs[2]: MoCo-v3 ResNet50 step (two views, base fwd+bwd x2, momentum fwd x2, EMA, InfoNCE,
    LARS; main_moco.py:321-370 defaults: dim 256, mlp 4096, T 1.0, m 0.99, lr 0.6 bs/256, wd 1e-6)."""
    from functools import partial
    import torch.distributed as dist
    from ssl4gie_amd import _lib, parallel
    from ssl4gie_amd.Models.moco_v3.moco import builder
    from ssl4gie_amd.Models.moco_v3.moco.optimizer import LARS
    from ssl4gie_amd.Models.resnet import resnet50
    _lib.load()
    rank, local, world = parallel.init_from_env()
    assert world == a.gpus
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    B = a.batch
    torch.manual_seed(0)
    model = builder.MoCo_ResNet(partial(resnet50, zero_init_residual=True), 256, 4096, 1.0)
    if world > 1:
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)  # main_moco.py:196
    model.to(dev).set_precision(a.precision)
    ddp = parallel.DataParallel(model) if world > 1 else None
    # lr: the reference ramps 0 -> 0.6 bs/256 over 10 warm-up epochs (main_moco.py:420-428); a
    # throughput run on uncorrelated noise views uses an early-warm-up value
    if a.optim == "arena":
        from ssl4gie_amd.optim import ArenaLARS
        opt = ArenaLARS(model, [p for p in model.parameters() if p.requires_grad],
                        lr=0.03 * B * world / 256, weight_decay=1e-6, momentum=0.9)
    else:
        opt = LARS([p for p in model.parameters() if p.requires_grad], lr=0.03 * B * world / 256,
                   weight_decay=1e-6, momentum=0.9)
    g = torch.Generator("cpu").manual_seed(rank)
    x1 = torch.randn(B, 3, 224, 224, generator=g).to(dev)
    x2 = torch.randn(B, 3, 224, 224, generator=g).to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = (ddp or model)(x1, x2, 0.99)
        loss.backward()   # under DataParallel the gradient exchange completes in here
        opt.step()
        return loss

    loss, dt, verify = timed_steps(step, a, world, dev, ddp)
    roof = roofline_pass(step, a, a.workload)
    if world > 1:
        dist.barrier()
    if rank == 0:
        ips = B * world * a.steps / dt
        print(json.dumps({**dp_info(ddp, a.steps + a.warmup, verify), 
            "metric": "image pairs/sec (fwd+bwd+LARS) MoCo-v3 ResNet50 224x224 (BASELINE.json configs[2])",
            "value": round(ips, 1), "unit": "image pairs/sec", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
            "median_ms_per_step": a.median_ms_per_step, "host_enqueue_ms": a.host_enqueue_ms, "roofline": roof, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
            "config": {"workload": "MoCo_ResNet(resnet50, 256, 4096, T=1.0), m=0.99, two synthetic views "
                                   "resident in HBM, LARS",
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}"},
            "model_mfma_frac": round(ips / world * 65.57 / 1e3 / PEAK_BF16_TFLOPS, 4),
            "final_loss": round(float(loss.detach()), 5)}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    dp_exit_if_broken(verify, world)


def bench_vit(a):
    """Plain ViT-B fwd + bwd + AdamW at bs 256 (linear-classifier finetune of a MAE trunk,
    `get_MAE_backbone(None, True, 6, False, None, False)`; train_classification.py): the un-masked
    N = 197 trunk that the north_star's "ViT-B fwd+bwd at bs=256/GPU" target is phrased on."""
    import torch.distributed as dist
    from ssl4gie_amd import _lib, parallel
    from ssl4gie_amd.Models import models
    _lib.load()
    rank, local, world = parallel.init_from_env()
    assert world == a.gpus
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    B = a.batch
    torch.manual_seed(0)
    model = models.ViT_from_MAE(None, True, 6, False, None, False, None, 768, 12, 12, "cls")
    model.to(dev).set_precision(a.precision)
    ddp = parallel.DataParallel(model) if world > 1 else None
    if a.optim == "arena":
        from ssl4gie_amd.optim import ArenaAdamW
        opt = ArenaAdamW(model, [p for p in model.parameters() if p.requires_grad], lr=1e-4)
    else:
        opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4, fused=True)
    g = torch.Generator("cpu").manual_seed(rank)
    imgs = torch.randn(B, 3, 224, 224, generator=g).to(dev)
    labels = torch.randint(0, 6, (B,), generator=g).to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy((ddp or model)(imgs), labels)
        loss.backward()   # under DataParallel the gradient exchange completes in here
        opt.step()
        return loss

    loss, dt, verify = timed_steps(step, a, world, dev, ddp)
    roof = roofline_pass(step, a, a.workload)
    if world > 1:
        dist.barrier()
    if rank == 0:
        ips = B * world * a.steps / dt
        print(json.dumps({**dp_info(ddp, a.steps + a.warmup, verify), 
            "metric": "images/sec (fwd+bwd+AdamW) ViT-B 224x224 linear-head finetune, un-masked trunk",
            "value": round(ips, 1), "unit": "images/sec", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
            "median_ms_per_step": a.median_ms_per_step, "host_enqueue_ms": a.host_enqueue_ms, "roofline": roof, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
            "config": {"workload": "ViT_from_MAE(head=True, num_classes=6) + cross entropy + AdamW(1e-4), "
                                   "synthetic images resident in HBM",
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}"},
            "model_mfma_frac": round(ips / world * 105.38 / 1e3 / PEAK_BF16_TFLOPS, 4),
            "final_loss": round(float(loss.detach()), 5)}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    dp_exit_if_broken(verify, world)


def bench_det(a):
    """detection ViT-B backbone (windowed + global attention, N = 4096) + ViTDet
    FPN at 1024 x 1024, fwd + bwd + AdamW with a synthetic quadratic loss on the five pyramid maps
    (RPN / RoI heads are torchvision Python in the reference and out of scope)."""
    import torch.distributed as dist
    from ssl4gie_amd import _lib, parallel
    from ssl4gie_amd.Models import models
    _lib.load()
    rank, local, world = parallel.init_from_env()
    assert world == a.gpus
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    B = a.batch if a.batch != 256 else 4
    torch.manual_seed(0)
    model = models.VisionTransformer_from_Any(False, 0, False, None, True, 1024, 768, 12, 12, "cls")
    model.to(dev).set_precision(a.precision)
    ddp = parallel.DataParallel(model) if world > 1 else None
    if a.optim == "arena":
        from ssl4gie_amd.optim import ArenaAdamW
        opt = ArenaAdamW(model, [p for p in model.parameters() if p.requires_grad], lr=1e-4)
    else:
        opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4, fused=True)
    g = torch.Generator("cpu").manual_seed(rank)
    imgs = torch.randn(B, 3, 1024, 1024, generator=g).to(dev)

    def step():
        opt.zero_grad(set_to_none=True)
        out = (ddp or model)(imgs)
        loss = sum((v * v).mean() for v in out.values())
        loss.backward()   # under DataParallel the gradient exchange completes in here
        opt.step()
        return loss

    loss, dt, verify = timed_steps(step, a, world, dev, ddp)
    roof = roofline_pass(step, a, a.workload)
    if world > 1:
        dist.barrier()
    if rank == 0:
        ips = B * world * a.steps / dt
        # GEMM-shaped MACs per image: 12 blocks x 4096 tokens x 12 D^2, windowed attention
        # 8 x 2 N 256 D, global attention 4 x 2 N^2 D, patch embed, pyramid convolutions
        D, N = 768, 4096
        gmac = (12 * N * 12 * D * D + 8 * 2 * N * 256 * D + 4 * 2 * N * N * D + N * D * D
                + 2 * (N * D * 4 * D) + 2 * (4 * N * D * 4 * D) + (N // 4 + N + 4 * N + 16 * N) * D * 256
                + (N // 4 + N + 4 * N + 16 * N) * 9 * 256 * 256) / 1e9
        print(json.dumps({**dp_info(ddp, a.steps + a.warmup, verify), 
            "metric": "images/sec (fwd+bwd+AdamW) detection ViT-B backbone + ViTDet FPN 1024x1024 (SURVEY 8f-1)",
            "value": round(ips, 2), "unit": "images/sec", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
            "median_ms_per_step": a.median_ms_per_step, "host_enqueue_ms": a.host_enqueue_ms, "roofline": roof, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
            "config": {"workload": "VisionTransformer_from_Any(det=True, fixed_size=1024) + ViTDet_FPN, "
                                   "synthetic images resident in HBM, quadratic loss on the pyramid maps",
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}"},
            "gflop_per_image_fwd_bwd": round(6 * gmac, 1),
            "model_mfma_frac": round(ips / world * 6 * gmac / 1e3 / PEAK_BF16_TFLOPS, 4),
            "final_loss": round(float(loss.detach()), 5)}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    dp_exit_if_broken(verify, world)


def bench_bt(a):
    """configs[4]: Barlow Twins on a ViT-B trunk, two views of 512 images per GPU, projector
    8192-8192-8192, lambda 0.0051, LARS (the build's own specification of the published method —
    the reference has no Barlow Twins trainer); the exchange is the all-reduce of
    the 8192 x 8192 cross-correlation."""
    import torch.distributed as dist
    from ssl4gie_amd import _lib, parallel
    from ssl4gie_amd.Models.barlow_twins import BarlowTwins
    from ssl4gie_amd.Models.moco_v3 import vits
    from ssl4gie_amd.Models.moco_v3.moco.optimizer import LARS
    _lib.load()
    rank, local, world = parallel.init_from_env()
    assert world == a.gpus
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    B = a.batch if a.batch != 256 else 512
    torch.manual_seed(0)
    bb = vits.vit_base(num_classes=8)
    del bb.head
    model = BarlowTwins(bb, 768, "8192-8192-8192", lambd=0.0051)
    if world > 1:
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    model.to(dev).set_precision(a.precision)
    ddp = parallel.DataParallel(model) if world > 1 else None
    if a.optim == "arena":  # the same update as one kernel over the parameter arena (as bench_moco)
        from ssl4gie_amd.optim import ArenaLARS
        opt = ArenaLARS(model, [p for p in model.parameters() if p.requires_grad], lr=0.02 * B * world / 256,
                        weight_decay=1.5e-6, momentum=0.9)
    else:
        opt = LARS([p for p in model.parameters() if p.requires_grad], lr=0.02 * B * world / 256,
                   weight_decay=1.5e-6, momentum=0.9)
    g = torch.Generator("cpu").manual_seed(rank)
    y1 = torch.randn(B, 3, 224, 224, generator=g).to(dev)
    y2 = (y1.cpu() + 0.5 * torch.randn(B, 3, 224, 224, generator=g)).to(dev)  # correlated views

    def step():
        opt.zero_grad(set_to_none=True)
        loss = (ddp or model)(y1, y2)
        loss.backward()   # under DataParallel the gradient exchange completes in here
        opt.step()
        return loss

    loss, dt, verify = timed_steps(step, a, world, dev, ddp)
    roof = roofline_pass(step, a, a.workload)
    if world > 1:
        dist.barrier()
    if rank == 0:
        ips = B * world * a.steps / dt
        # 212.44 GFLOP per image (two views, trunk + projector, fwd+bwd) + the three
        # 8192 x 8192 x B cross-correlation products (c, dz_A, dz_B) = 6 * 8192^2 FLOP per image
        gflop_img = 212.44 + 6 * 8192 * 8192 / 1e9
        print(json.dumps({**dp_info(ddp, a.steps + a.warmup, verify), 
            "metric": "images/sec (two views, fwd+bwd+LARS) Barlow Twins ViT-B 224x224 (BASELINE.json configs[4])",
            "value": round(ips, 1), "unit": "images/sec", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
            "median_ms_per_step": a.median_ms_per_step, "host_enqueue_ms": a.host_enqueue_ms, "roofline": roof, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
            "config": {"workload": "BarlowTwins(ViT-B trunk, projector 8192-8192-8192, lambda 0.0051), two "
                                   "synthetic views resident in HBM, LARS; own specification (absent "
                                   "from the reference)",
                       "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}"},
            "model_mfma_frac": round(ips / world * gflop_img / 1e3 / PEAK_BF16_TFLOPS, 4),
            "final_loss": round(float(loss.detach()), 5)}), flush=True)
    if world > 1:
        dist.destroy_process_group()
    dp_exit_if_broken(verify, world)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--prof-steps", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--optim", default="arena", choices=["torch", "arena"],
                    help="arena: the optimizer update as kernels over the parameter arena (ssl4gie_amd.optim: "
                         "torch.optim.AdamW's / the reference LARS's arithmetic, tests/test_gpu_optim.py; "
                         "AdamW also emits the bf16 operand copies) — measured faster than torch's fused "
                         "step once that refreshes the operand copies too; torch: torch.optim fused / "
                         "foreach steps")
    ap.add_argument("--workload", default="mae", choices=["mae", "depth", "moco", "bt", "det", "vit"],
                    help="mae = BASELINE.json configs[1] (the headline metric); depth = configs[3] "
                         "(ViT-B + DPT depth finetune step, bs 128/GPU) as an extra measurement")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="launch check without a GPU: the ranks meet (gloo), sum their ranks and rank 0 "
                         "prints one JSON line — what tests/test_abi_and_host.py runs through the self-launcher")
    a = ap.parse_args()
    if a.rendezvous_only:
        import torch.distributed as dist
        from ssl4gie_amd import parallel
        rank, local, world = parallel.init_from_env("gloo")
        assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
        t = torch.tensor([float(rank)])
        if world > 1:
            dist.all_reduce(t)
        out = {"rendezvous": world, "rank_sum": float(t)}
        verify = None
        if world > 1:
            # the N > 1 self-check of the real workloads (dp_verify), on a CPU toy: a small arena model under
            # DataParallel + SGD, different data per rank
            import torch.nn as nn
            from ssl4gie_amd.engine import GradSink, ParamArena

            class Toy(nn.Module):
                def __init__(self):
                    super().__init__()
                    self.a, self.b = nn.Linear(8, 16), nn.Linear(16, 4)
                    self._a = ParamArena(list(self.parameters()))
                    self._s = GradSink(self._a)

                def arena(self):
                    return self._a

                def sink(self):
                    return self._s

                def forward(self, x):
                    return (self.b(torch.tanh(self.a(x))) ** 2).mean()

            torch.manual_seed(100 + rank)
            toy = Toy()
            ddp = parallel.DataParallel(toy)
            opt = torch.optim.SGD(toy.parameters(), lr=0.1)
            x = torch.randn(5, 8, generator=torch.Generator().manual_seed(7 + rank))

            def step():
                opt.zero_grad(set_to_none=True)
                loss = ddp(x)
                loss.backward()
                opt.step()
                return loss

            _, _, verify = timed_steps(step, a, world, torch.device("cpu"), ddp)
            out.update(dp_info(ddp, a.steps + a.warmup, verify))
        if rank == 0:
            print(json.dumps(out), flush=True)
        if world > 1:
            dist.destroy_process_group()
        dp_exit_if_broken(verify, world)
        return
    if a.workload == "depth":
        return bench_depth(a)
    if a.workload == "bt":
        return bench_bt(a)
    if a.workload == "det":
        return bench_det(a)
    if a.workload == "vit":
        return bench_vit(a)
    if a.workload == "moco":
        return bench_moco(a)

    import torch.distributed as dist
    from ssl4gie_amd import _lib, parallel
    from ssl4gie_amd.Models.mae import models_mae

    _lib.load()  # the HIP extension is mandatory
    rank, local, world = parallel.init_from_env()
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    torch.manual_seed(0)
    model = models_mae.mae_vit_base_patch16(norm_pix_loss=True).to(dev).set_precision(a.precision)
    ddp = parallel.DataParallel(model) if world > 1 else None
    if a.optim == "arena":  # the same update as one kernel over the parameter arena
        from ssl4gie_amd.optim import ArenaAdamW
        opt = ArenaAdamW(model, param_groups(model), lr=1.5e-4, betas=(0.9, 0.95),
                         overlap_backward=os.environ.get("SSL4GIE_OPT_OVERLAP", "0") == "1")  # measured: no gain
    else:
        try:
            opt = torch.optim.AdamW(param_groups(model), lr=1.5e-4, betas=(0.9, 0.95), fused=True)
        except Exception:
            opt = torch.optim.AdamW(param_groups(model), lr=1.5e-4, betas=(0.9, 0.95))

    # seed + rank (main_pretrain.py:116); SSL4GIE_BENCH_SAME_DATA=1 feeds every rank the same batch,
    # which makes the N-rank run numerically comparable to N=1 (tools/gpu_rehearse_dp.sh)
    same = os.environ.get("SSL4GIE_BENCH_SAME_DATA") == "1"
    g = torch.Generator("cpu").manual_seed(0 if same else rank)
    imgs = torch.randn(a.batch, 3, 224, 224, generator=g).pin_memory().to(dev, non_blocking=True)

    def step():
        opt.zero_grad(set_to_none=True)
        loss, _, _ = (ddp or model)(imgs, mask_ratio=0.75)
        loss.backward()   # under DataParallel the gradient exchange completes in here
        opt.step()
        return loss

    loss, dt, verify = timed_steps(step, a, world, dev, ddp)
    final_loss = float(loss.detach())

    roof = roofline_pass(step, a, "mae")
    if world > 1:
        dist.barrier()

    if rank == 0:
        ips = a.batch * world * a.steps / dt
        line = {
            "metric": METRIC, "value": round(ips, 1), "unit": "images/sec", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
            "median_ms_per_step": a.median_ms_per_step, "host_enqueue_ms": a.host_enqueue_ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.precision, "data": "synthetic",
            "config": {"workload": "ViT-B MAE pretrain (mae_vit_base_patch16, norm_pix_loss, "
                                   "mask 0.75) fwd+bwd+grad-allreduce+AdamW, 224x224 synthetic "
                                   "N(0,1) images resident in HBM",
                       "batch_per_gpu": a.batch, "global_batch": a.batch * world,
                       "parallelism": f"dp{world}",
                       "optimizer": "AdamW(0.9,0.95) wd 0.05 (" + ("ssl4gie_amd.optim.ArenaAdamW" if a.optim == "arena" else "torch.optim.AdamW fused") + ")"},
            "images_per_sec_per_gpu": round(ips / world, 1),
            "model_mfma_frac": round(ips / world * GFLOP_PER_IMG / 1e3 / PEAK_BF16_TFLOPS, 4),
            "final_loss": round(final_loss, 5),
            "tree": tree_stamp(),
        }
        line.update(dp_info(ddp, a.steps + a.warmup, verify))
        if roof is not None:
            line["roofline"] = roof
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()
    dp_exit_if_broken(verify, world)


if __name__ == "__main__":
    rc = self_launch(sys.argv[1:])
    if rc is not None:
        sys.exit(rc)
    main()
