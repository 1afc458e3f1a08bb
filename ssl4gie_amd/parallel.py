"""One-process-per-GPU data parallelism over RCCL/xGMI (torch.distributed backend "nccl" on ROCm;
"gloo" for the CPU tests).

Reference behaviour replaced: `DistributedDataParallel(model,
find_unused_parameters=True)` gradient averaging (Depth_estimation/train_depth.py:226-229,
Models/mae/main_pretrain.py:175), the per-step `dist.all_reduce(loss)` + `/world_size`
(train_depth.py:47-48) and the rank-0 parameter broadcast DDP performs at construction.

Design (MI355X-first, not torch DDP's reducer):
  * gradients already live in ONE flat fp32 arena (engine.ParamArena) laid out in registration
    order, and backward fills it strictly from the end towards the start (decoder_pred ...
    patch_embed).  A bucket is therefore just a contiguous slice [lo, hi) of the arena: no
    flatten/unflatten copies, no per-parameter hooks;
  * the block executor calls back after each transformer block's gradients are enqueued; once
    >= bucket_bytes of arena have become final, the slice is all-reduced on a dedicated comm
    stream that waits on an event of the compute stream — communication overlaps the rest of
    backward.  Buckets are large (default 64 MiB): xGMI is point-to-point, few big collectives beat
    many small ones;
  * parameters that never receive a gradient (the reference needs find_unused_parameters=True for
    `norm.*` in dense mode) are simply zeros in the arena: nothing waits for them;
  * averaging (1/world) is folded into the collective when the backend supports ReduceOp.AVG,
    otherwise one scale kernel per bucket.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn


class DataParallel:
    """Wraps an EngineModule-like model (anything exposing `.arena()` -> ParamArena-like object with
    `.data`, `.grad`, `.span(params)` and an assignable `_grad_hook`)."""

    def __init__(self, model, process_group=None, bucket_bytes: int = 64 << 20,
                 overlap: bool = True, broadcast_parameters: bool = True):
        assert dist.is_initialized(), "init_process_group first (one process per GPU)"
        self.module = model
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.overlap = overlap
        self._arena = model.arena()
        self._is_cuda = self._arena.grad.is_cuda
        self._comm_stream = torch.cuda.Stream() if self._is_cuda else None
        self._handles: List = []
        self._hi = self._arena.grad.numel()  # everything in [self._hi, numel) already reduced
        self._avg_op = self._pick_avg_op()
        self.n_collectives = 0
        if self._is_cuda and self.world > 1:
            # the 256x256 GEMM workgroups own a CU's whole LDS, so RCCL's kernels need CUs of their
            # own while a bucket is in flight: size the persistent GEMM grids for 256 - R CUs
            # (init_from_env caps RCCL at R channels).  SSL4GIE_COMM_CUS=0 disables the reservation.
            from . import _lib
            r = comm_cus()
            if r > 0:
                _lib.check(_lib.load().ssl4gie_set_compute_cus(256 - r), "set_compute_cus")
        if broadcast_parameters and self.world > 1:
            dist.broadcast(self._arena.data, src=0, group=process_group)
            try:  # the flat buffer was written, not the parameter views: refresh operand caches
                from .engine import bump_weights_epoch
                bump_weights_epoch()
            except ImportError:  # toy models of the CPU tests do not load the engine
                pass
        model._grad_hook = self._on_module_grads if overlap else None

    # the model is used exactly like the wrapped module
    def __call__(self, *a, **k):
        self._hi = self._arena.grad.numel()
        return self.module(*a, **k)

    def parameters(self):
        return self.module.parameters()

    def state_dict(self, *a, **k):
        return self.module.state_dict(*a, **k)

    def _pick_avg_op(self):
        try:
            backend = dist.get_backend(self.pg)
        except Exception:
            backend = "gloo"
        return dist.ReduceOp.AVG if backend == "nccl" else None

    # ---------------------------------------------------------------- bucket machinery
    def _reduce_slice(self, lo: int, hi: int):
        if hi <= lo or self.world == 1:
            return
        g = self._arena.grad[lo:hi]
        self.n_collectives += 1
        if self._is_cuda:
            self._comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._comm_stream):
                if self._avg_op is not None:
                    h = dist.all_reduce(g, op=self._avg_op, group=self.pg, async_op=True)
                else:
                    h = dist.all_reduce(g, group=self.pg, async_op=True)
                self._handles.append((h, g))
        else:
            h = dist.all_reduce(g, group=self.pg, async_op=True)
            self._handles.append((h, g))

    def _on_module_grads(self, module: nn.Module):
        """Engine callback: the gradients of `module` (a Block) and of everything registered after
        it are enqueued on the compute stream."""
        ps = [p for p in module.parameters()]
        if not ps:
            return
        lo, _ = self._arena.span(ps)
        if self._hi - lo >= self.bucket_elems:
            self._reduce_slice(lo, self._hi)
            self._hi = lo

    def finish(self):
        """Call after loss.backward() and before optimizer.step(): flushes the last bucket and makes
        the compute stream wait for all communication."""
        self._reduce_slice(0, self._hi)
        self._hi = 0
        for h, g in self._handles:
            h.wait()
            if self._avg_op is None and self.world > 1:
                g.div_(self.world)
        self._handles.clear()
        if self._is_cuda:
            torch.cuda.current_stream().wait_stream(self._comm_stream)
        self._hi = self._arena.grad.numel()

    # ---------------------------------------------------------------- small collectives
    def all_reduce_mean(self, t: torch.Tensor) -> torch.Tensor:
        """Mean over ranks of a small tensor (logged loss; reference train_depth.py:47-48)."""
        if self.world == 1:
            return t
        t = t.detach().clone()
        dist.all_reduce(t, group=self.pg)
        return t / self.world


def comm_cus() -> int:
    """CUs left to the communication kernels in data-parallel runs (SSL4GIE_COMM_CUS, default 32:
    the MAE step time is flat between 208 and 256 compute CUs, profiles/r01o_*)"""
    import os
    return max(0, min(128, int(os.environ.get("SSL4GIE_COMM_CUS", "32"))))


def init_from_env(backend: Optional[str] = None):
    """Initialise torch.distributed from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT).  Returns (rank, local_rank, world)."""
    import os
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        # rehearsal on a box with fewer GPUs than ranks (SSL4GIE_DIST_BACKEND=gloo): ranks share devices
        local = local % torch.cuda.device_count()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("SSL4GIE_DIST_BACKEND") or \
                ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            if comm_cus() > 0:  # one RCCL channel = one workgroup = one CU (see DataParallel)
                os.environ.setdefault("NCCL_MAX_NCHANNELS", str(comm_cus()))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world
