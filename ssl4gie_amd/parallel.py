"""One-process-per-GPU data parallelism over RCCL/xGMI (torch.distributed backend "nccl" on ROCm;
"gloo" for the CPU tests).

Reference behaviour replaced: `DistributedDataParallel(model,
find_unused_parameters=True)` gradient averaging (Depth_estimation/train_depth.py:226-229,
Models/mae/main_pretrain.py:175), the per-step `dist.all_reduce(loss)` + `/world_size`
(train_depth.py:47-48) and the rank-0 parameter broadcast DDP performs at construction.

Design (MI355X-first, not torch DDP's reducer):
  * gradients already live in ONE flat fp32 arena (engine.ParamArena) laid out in registration
    order, and backward fills it roughly from the end towards the start (decoder_pred ...
    patch_embed).  A bucket is therefore just a contiguous slice [lo, hi) of the arena: no
    flatten/unflatten copies, no per-bucket bookkeeping of parameter lists;
  * a post-accumulate-grad hook per trainable parameter reports when its gradient is final (all
    uses of the parameter in this backward have run); once >= bucket_bytes of arena above the
    frontier have become final, the slice is all-reduced on a dedicated comm stream that waits on
    an event of the compute stream — communication overlaps the rest of backward for every model
    (transformer stacks, ResNet50 stages, DPT decoder, two-view wrappers).  Buckets are large
    (default 64 MiB): xGMI is point-to-point, few big collectives beat many small ones;
  * parameters that never receive a gradient (the reference needs find_unused_parameters=True for
    `norm.*` in dense mode) are simply zeros in the arena: nothing waits for them;
  * averaging (1/world) is folded into the collective when the backend supports ReduceOp.AVG,
    otherwise one scale kernel per bucket.
"""
from __future__ import annotations

import weakref
from typing import List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn


_LIVE = weakref.WeakSet()  # wrappers whose communication an optimizer step must wait for


def before_optimizer_step():
    """Second line of defence behind the end-of-backward call-back: every optimizer step (torch
    optimizers through a global pre-step hook, ssl4gie_amd.optim through a direct call) first closes
    any pass a wrapper still has open and makes the compute stream wait for the comm stream — an
    optimizer can never step on un-reduced gradients, whatever the training loop looks like."""
    for dp in list(_LIVE):
        dp._before_step()


try:
    from torch.optim.optimizer import register_optimizer_step_pre_hook as _reg_pre_hook
    _reg_pre_hook(lambda _opt, _args, _kwargs: before_optimizer_step())
except ImportError:  # torch < 2.0: ssl4gie_amd.optim still calls before_optimizer_step() itself
    pass


class DataParallel(nn.Module):
    """Drop-in for `torch.nn.parallel.DistributedDataParallel` on an EngineModule-like model (anything
    exposing `.arena()` -> ParamArena-like object with `.data`, `.grad`, `.params`, `.span(params)`,
    `.grad_view(p)`): replace the class name in the reference's train loop, nothing else.

        model = DataParallel(model, device_ids=[gpu], find_unused_parameters=True)  # train_depth.py:226-229
        model.train(); ...; loss.backward(); optimizer.step()                        # :35-48, unchanged

    It is an `nn.Module` holding the model as `self.module`: `train()` / `eval()` / `load_state_dict()` /
    `named_parameters()` work on the wrapper, `state_dict()` carries the `module.` prefix the reference's
    MoCo checkpoints have (`main_moco.py:313`, consumed by `convert_to_deit.py:24-32`), and
    `model.module.state_dict()` is the un-prefixed dict `train_depth.py:357` saves.  The gradient
    exchange finishes INSIDE `backward()` (an autograd-engine call-back queued by the first hook of a
    pass, as torch DDP's reducer does), so `scaler.scale(loss).backward()` is followed directly by
    `scaler.step(optimizer)`; `finish()` remains as an idempotent no-op for older callers.

    Readiness is per PARAMETER and comes from autograd itself: a post-accumulate-grad hook on every
    trainable parameter fires once per backward, after the LAST node that uses the parameter has
    enqueued its kernels (autograd runs a leaf's AccumulateGrad only when every use has delivered
    its contribution) — so a parameter used twice in one step (both views of MoCo / Barlow Twins),
    a parameter of an adopted child module, a convolution or BatchNorm parameter are all handled by
    the same rule.  The arena is walked from its end: the "frontier" is the lowest offset above which
    every trainable parameter is final (or is known never to receive a gradient: learnt during the
    first pass and agreed between the ranks, the reference's `find_unused_parameters=True`); whenever
    `bucket_bytes` of final gradients have accumulated above the frontier, that contiguous slice is
    all-reduced on the comm stream.  The slices are a PLAN: a pure function of the arena layout, the
    bucket size and the agreed unused set, so every rank issues the same collectives of the same sizes
    in the same order whatever its own graph did in this pass (a parameter whose hook does not fire on
    one rank only delays that rank's slices to the end of its backward; it never changes them).  Runs
    of parameters learnt as unused are left out of the overlapped slices and go out as "tail" slices
    at the end of every pass — a few KB to MB of zeros normally — so one that does receive a gradient
    later (the graph changed, on any subset of the ranks) is still averaged in that same pass with
    nothing in flight over it; runs of frozen parameters (MoCo's momentum encoder: half the arena)
    split the arena into segments that are never communicated."""

    FROZEN_GAP_ELEMS = 1 << 18  # a frozen run >= 1 MiB ends a segment (not worth carrying along)

    def __init__(self, module, device_ids=None, output_device=None, find_unused_parameters=True,
                 process_group=None, bucket_bytes: int = 64 << 20, overlap: bool = True,
                 broadcast_parameters: bool = True, broadcast_buffers: bool = True, **_ddp_kwargs):
        """`device_ids`, `output_device`, `find_unused_parameters`, `broadcast_buffers` and further
        keyword arguments of torch's DDP are accepted so that the reference's constructor calls
        (`main_pretrain.py:175`, `train_depth.py:226-229`, `main_moco.py:208`) stay as they are:
        one process drives one device (the module's), unused parameters are always tolerated."""
        super().__init__()
        assert dist.is_initialized(), "init_process_group first (one process per GPU)"
        self.module = module
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.overlap = overlap
        self.require_backward_grad_sync = True   # torch DDP's name; False inside no_sync()
        self._arena = None
        self._hooks: List = []
        self._handles: List = []
        self._step = 0
        self._armed = False      # the end-of-backward call-back of the current pass is queued
        self.n_collectives = 0
        self.n_late = 0  # gradients seen on parameters learnt as unused (carried by the tail slices)
        self.n_passes = 0        # backward passes closed (from inside backward or by finish())
        self.n_overlapped = 0    # collectives that left while backward was still running
        self._n_at_pass_start = 0
        self._bind(module.arena())
        self._is_cuda = self._arena.grad.is_cuda
        self._comm_stream = torch.cuda.Stream() if self._is_cuda else None
        self._avg_op = self._pick_avg_op()
        # transport of the bucket all-reduce: RCCL (default) or the library's direct xGMI exchange
        self._direct = None
        self.compute_cus = None  # CUs the persistent GEMM grids are sized for (set below when CUs are reserved)
        # SSL4GIE_ALLREDUCE = rccl (default) | direct | auto.  `auto` proves the choice on this node before the first
        # step: a 64-MiB slice goes through both transports, the direct result must be bitwise equal on every rank
        # and within 1e-6 of RCCL's, and direct is taken only if it is also faster (probe_transports; the verdict
        # is agreed between the ranks, a failing direct path falls back in-process).  `transport_probe` keeps the
        # record (bench.py prints it in the N > 1 lines).
        self.transport_probe = None
        import os
        mode = os.environ.get("SSL4GIE_ALLREDUCE", "rccl").lower()
        if self._is_cuda and self.world > 1 and mode in ("direct", "auto"):
            max_elems = min(self.bucket_elems * 2, self._arena.grad.numel())
            if mode == "direct":
                self._direct = DirectAllReduce(max_elems, process_group)
            else:
                fault = os.environ.get("SSL4GIE_AR_PROBE_FAULT")   # test hook: "<rank>" corrupts that rank's direct result
                self._direct, self.transport_probe = probe_transports(
                    process_group, self._arena.grad.device, lambda: DirectAllReduce(max_elems, process_group),
                    n_elems=min(16 << 20, max_elems), fault_rank=int(fault) if fault not in (None, "") else None)
        if self._is_cuda and self.world > 1:
            # the 256x256 GEMM workgroups own a CU's whole LDS, so RCCL's kernels need CUs of their
            # own while a bucket is in flight: size the persistent GEMM grids for 256 - R CUs
            # (init_from_env caps RCCL at R channels).  SSL4GIE_COMM_CUS=0 disables the reservation.
            from . import _lib
            r = comm_cus()
            if r > 0:
                _lib.check(_lib.load().ssl4gie_set_compute_cus(256 - r), "set_compute_cus")
                self.compute_cus = 256 - r
        if broadcast_parameters and self.world > 1:
            dist.broadcast(self._arena.data, src=0, group=process_group)
            # DDP also broadcasts buffers (BatchNorm running statistics) from rank 0
            for b in (module.buffers() if (broadcast_buffers and hasattr(module, "buffers")) else ()):
                if b.numel():
                    dist.broadcast(b, src=0, group=process_group)
            try:  # the flat buffer was written, not the parameter views: refresh operand caches
                from .engine import bump_weights_epoch
                bump_weights_epoch()
            except ImportError:  # toy models of the CPU tests do not load the engine
                pass
        _LIVE.add(self)

    @property
    def transport(self) -> str:
        """what carries the gradient slices: "direct" (csrc/allreduce.hip), else the
        torch.distributed backend's name ("nccl" is RCCL on ROCm; "gloo" in CPU tests / rehearsals)"""
        if self._direct is not None:
            return "direct"
        try:
            b = str(dist.get_backend(self.pg))
        except Exception:
            b = "unknown"
        return "rccl" if b == "nccl" else b

    # ---------------------------------------------------------------- arena index
    def _bind(self, arena):
        """(Re)build the per-parameter index for `arena` and hang the readiness hooks."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self._arena = arena
        spans = []
        for p in arena.params:
            lo, hi = arena.span([p])
            spans.append((lo, hi, p))
        spans.sort(key=lambda t: -t[0])  # walk order: from the end of the arena
        # segments = maximal runs of trainable parameters (small frozen gaps are carried along)
        self._items = []     # trainable only: [lo, hi, param, segment]
        self._seg_hi = []    # per segment: end offset
        seg, gap, prev_lo = -1, self.FROZEN_GAP_ELEMS, None
        for lo, hi, p in spans:
            if not p.requires_grad:
                gap += hi - lo
                continue
            if gap >= self.FROZEN_GAP_ELEMS or prev_lo is None:
                seg += 1
                self._seg_hi.append(hi)
            gap, prev_lo = 0, lo
            self._items.append((lo, hi, p, seg))
        self._index = {id(p): i for i, (_, _, p, _) in enumerate(self._items)}
        n = len(self._items)
        self._stamp = [0] * n        # pass in which the parameter's hook last fired
        self._unused = [False] * n   # learnt: never receives a gradient
        self._learnt = False
        self._uses = [0] * n         # uses reported by the engine (GradSink.tracker) in this pass
        self._expected = [0] * n     # learnt: uses per backward pass (2 for two-view models)
        sink = getattr(self.module, "sink", None)
        if callable(sink) and self.overlap and self.world > 1:
            sink().tracker = self._on_use_done
        self._build_plan()
        self._reset_pass()
        if self.world > 1:
            try:   # engine.wgrad_fork keeps its side stream beside THESE hooks (the comm stream waits on it)
                from .engine import SAFE_POST_ACC_HOOK_IDS as safe_ids
            except ImportError:  # toy models of the CPU tests do not load the engine
                safe_ids = set()
            for _, _, p, _ in self._items:
                h = p.register_post_accumulate_grad_hook(self._on_param)
                safe_ids.add(h.id)
                self._hooks.append(h)

    def _build_plan(self):
        """The slices of a pass as a pure function of (arena layout, bucket size, unused set): items are
        walked from the end of the arena; a run of consecutive items of one segment and one kind (used /
        learnt-unused) is cut greedily into slices of >= bucket_elems (its remainder is a slice of its
        own).  `_plan` holds the slices of the used runs in walk order as (lo, hi, end_item): slice s may
        leave once items [0, end_item) are final and slices [0, s) have left; `_tails` the slices of the
        unused runs, sent at the end of every pass.  Every rank builds the same plan from the same agreed
        unused set, so collective counts, sizes and order never depend on a rank's own graph."""
        items, unused, n = self._items, self._unused, len(self._items)
        plan, tails, i = [], [], 0
        while i < n:
            seg, u, j = items[i][3], unused[i], i
            while j < n and items[j][3] == seg and unused[j] == u:
                j += 1
            first = i == 0 or items[i - 1][3] != seg
            hi = self._seg_hi[seg] if first else items[i - 1][0]
            if u:
                tails.append((items[j - 1][0], hi))
            else:
                for k in range(i, j):
                    lo = items[k][0]
                    if hi - lo >= self.bucket_elems or k == j - 1:
                        plan.append((lo, hi, k + 1))
                        hi = lo
            i = j
        self._plan, self._tails = plan, tails

    def _reset_pass(self):
        self._k = 0                  # items [0, k) are final
        self._sent = 0               # plan slices [0, sent) have been handed to a collective
        self._fired = self._late = 0
        self._uses = [0] * len(self._items) if hasattr(self, "_items") else []
        self._armed = False
        self._step += 1

    # the wrapper is used exactly like the wrapped module
    def forward(self, *a, **k):
        arena = self.module.arena()
        if arena is not self._arena:  # .to() / a new parameter rebuilt the arena
            self._drain()
            self._bind(arena)
        elif self._armed or self._fired:
            # a backward pass that raised half-way never reached its call-back: start afresh
            self._drain()
            self._reset_pass()
        return self.module(*a, **k)

    def arena(self):
        return self.module.arena()

    def sink(self):
        return self.module.sink()

    class _NoSync:
        def __init__(self, dp):
            self.dp = dp

        def __enter__(self):
            self.prev = self.dp.require_backward_grad_sync
            self.dp.require_backward_grad_sync = False

        def __exit__(self, *exc):
            self.dp.require_backward_grad_sync = self.prev

    def no_sync(self):
        """torch DDP's gradient-accumulation context: backward passes inside it add into the arena
        without communicating; the first pass outside it averages the accumulated gradients"""
        return DataParallel._NoSync(self)

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
            self._wait_wgrad_stream()
            if self._direct is not None:
                self._direct.all_reduce_(g, 1.0 / self.world, self._comm_stream)
                return
            with torch.cuda.stream(self._comm_stream):
                if self._avg_op is not None:
                    h = dist.all_reduce(g, op=self._avg_op, group=self.pg, async_op=True)
                else:
                    h = dist.all_reduce(g, group=self.pg, async_op=True)
                self._handles.append((h, g))
        else:
            h = dist.all_reduce(g, group=self.pg, async_op=True)
            self._handles.append((h, g))

    def _wait_wgrad_stream(self):
        """gradients of deferred weight-gradient groups are produced on the library's side stream
        (ssl4gie_wgrad_group): the comm stream waits for the latest group of either slot"""
        try:
            from . import _lib
            L = _lib.load()
        except Exception:
            return
        cs = self._comm_stream.cuda_stream
        for slot in (0, 1):
            L.ssl4gie_wgrad_wait(slot, cs)
        # ... and the single layers' weight gradients (engine.wgrad_fork: LinearFn / Conv3x3Fn) on theirs: whatever
        # has been enqueued there so far belongs to gradients whose hooks have fired
        from .engine import _WG
        if _WG["stream"] is not None:
            self._comm_stream.wait_stream(_WG["stream"])

    def _adopt(self, p):
        """a gradient autograd produced with torch ops (outside the engine's sinks) is moved into the
        parameter's arena slice, so that it is communicated and the arena optimizers see it"""
        v = self._arena.grad_view(p)
        if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
            v.copy_(p.grad)
            p.grad = v

    def _send(self, upto: int, force: bool = False):
        """hand the plan's slices to collectives, in plan order: those whose items [.., end_item) are all
        final (end_item <= upto), or — `force`, the end of the pass — every slice that has not left yet,
        followed by the tail slices of the learnt-unused runs."""
        plan = self._plan
        while self._sent < len(plan) and (force or plan[self._sent][2] <= upto):
            lo, hi, _ = plan[self._sent]
            self._reduce_slice(lo, hi)
            self._sent += 1
        if force:
            for lo, hi in self._tails:
                self._reduce_slice(lo, hi)

    def _arm(self):
        """queue the end-of-backward call-back (once per pass; only possible from inside backward)"""
        if not self._armed:
            self._armed = True
            from torch.autograd import Variable
            Variable._execution_engine.queue_callback(self._finalize)

    def _on_use_done(self, params):
        """engine call-back (GradSink.tracker): the kernels of one use of `params` are enqueued.  A
        parameter whose learnt number of uses per pass has been reached is final NOW — before
        autograd gets round to its AccumulateGrad, which for a block stack (one autograd node for
        8-12 blocks) is only after the whole stack."""
        if not self.require_backward_grad_sync:
            return
        for p in params:
            i = self._index.get(id(p))
            if i is None:
                continue
            self._uses[i] += 1
            if self._learnt and self._expected[i]:
                if self._uses[i] == self._expected[i]:
                    self._on_param(p, early=True)
                elif self._uses[i] > self._expected[i]:
                    raise RuntimeError(
                        "ssl4gie_amd.parallel.DataParallel: a parameter is used more often in this backward "
                        "pass than in the pass the communication schedule was learnt from (its gradient slice "
                        "may already be in flight); call ddp.relearn() after changing the graph")

    def relearn(self):
        """forget the learnt schedule (unused set, uses per pass): the next pass communicates at its
        end and learns again — call after changing which parameters take part in the graph"""
        self._drain()
        n = len(self._items)
        self._unused, self._expected, self._learnt = [False] * n, [0] * n, False
        self._build_plan()
        self._reset_pass()

    def _on_param(self, p, early=False):
        i = self._index.get(id(p))
        if i is None or self._stamp[i] == self._step:
            return
        self._arm()
        if not self.require_backward_grad_sync:
            return
        if not early:
            self._adopt(p)
        self._stamp[i] = self._step
        self._fired += 1
        if not self.overlap:
            return
        if self._unused[i]:
            # learnt as unused, used after all (the graph changed): its run is a tail slice, which every
            # rank sends at the end of every pass — nothing is in flight over it now, and it is averaged
            # in this same pass whether or not the other ranks saw the change
            self.n_late += 1
            self._late += 1
            return
        k, n = self._k, len(self._items)
        while k < n and (self._stamp[k] == self._step or self._unused[k]):
            k += 1
        if k != self._k:
            self._k = k
            self._send(k)

    def _on_module_grads(self, module: nn.Module):
        """Explicit readiness call for models that fill the arena without autograd leaves (kept for
        callers that drive the engine by hand): marks the module's parameters final."""
        for p in module.parameters():
            if p.requires_grad:
                self._on_param(p)

    def _agree_unused(self, used: List[bool]) -> List[bool]:
        """ranks agree on the unused set (a parameter used on ANY rank is waited for on all); called by
        every rank in the same pass (the first one after construction / relearn()), never on a
        rank-local condition"""
        t = torch.tensor([1.0 if u else 0.0 for u in used], dtype=torch.float32, device=self._arena.grad.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.pg)
        return [v == 0.0 for v in t.tolist()]

    def _finalize(self):
        """End of a backward pass (autograd-engine call-back; also reachable through finish() and the
        optimizer pre-step hook): send whatever has not gone out, then make the caller's stream wait
        for all communication.  The engine runs final call-backs on the stream that was current around
        `backward()`, after syncing it with the leaf streams."""
        self.n_overlapped += self.n_collectives - self._n_at_pass_start
        if self.world > 1 and self.require_backward_grad_sync:
            for _, _, p, _ in self._items:
                self._adopt(p)
            # the forced send covers items whose hook did not fire as well (zeros, or gradients of an
            # earlier pass that were never cleared): every rank sends the plan's slices either way
            self._send(len(self._items), force=True)
            if not self._learnt:
                # first pass after construction / relearn() — on every rank alike, so the agreement is a
                # collective all of them join: parameters whose hook fired on no rank do not take part in
                # this graph
                unused = self._agree_unused([s == self._step for s in self._stamp])
                if not all(unused):
                    self._unused = unused
                    self._expected, self._learnt = list(self._uses), True
                    self._build_plan()
                # else: a pass in which no hook fired on any rank (closed by finish() / an optimizer pre-step hook
                # over leftover handles) says nothing about the graph — learning from it would mark every
                # parameter unused and move all traffic to the un-overlapped tail for good; the next
                # real pass learns instead, on every rank alike (the agreement above is rank-uniform)
        self._drain()
        self.n_passes += 1
        self._n_at_pass_start = self.n_collectives
        self._reset_pass()

    def _drain(self):
        """wait for every collective in flight (host-side for gloo, stream-side for RCCL / direct)"""
        for h, g in self._handles:
            h.wait()
            if self._avg_op is None and self.world > 1:
                g.div_(self.world)
        self._handles.clear()
        if self._is_cuda:
            torch.cuda.current_stream().wait_stream(self._comm_stream)
            if self._direct is not None:
                self._direct.raise_if_failed()

    def _before_step(self):
        if self.world > 1 and (self._armed or self._fired or self._handles):
            self._finalize()
        elif self._is_cuda and self.world > 1:
            torch.cuda.current_stream().wait_stream(self._comm_stream)

    def finish(self):
        """No longer needed: the pass is closed from inside `backward()`.  Kept as an idempotent call
        for loops written against the round-2 interface — it closes a pass only if one is open."""
        if self.world > 1 and (self._armed or self._fired or self._handles):
            self._finalize()

    # ---------------------------------------------------------------- small collectives
    def all_reduce_mean(self, t: torch.Tensor) -> torch.Tensor:
        """Mean over ranks of a small tensor (logged loss; reference train_depth.py:47-48)."""
        if self.world == 1:
            return t
        t = t.detach().clone()
        dist.all_reduce(t, group=self.pg)
        return t / self.world


def choose_transport(records):
    """The rank-uniform verdict of a transport probe: `records` = one dict per rank with `direct_ok` (the direct
    path ran and matched the reference transport within tolerance), `checksum` (of the direct result: must be equal
    on every rank — replicas that disagree bit-wise drift apart), `ref_ms`, `direct_ms`.  -> (chosen, reason) with
    chosen in {"direct", "rccl"}; a pure function, every rank evaluates it on the same gathered list."""
    if not all(r.get("direct_ok") for r in records):
        bad = [i for i, r in enumerate(records) if not r.get("direct_ok")]
        return "rccl", f"direct transport failed or mismatched on rank(s) {bad}: " + \
            "; ".join(str(records[i].get("error", "mismatch")) for i in bad)
    if len({r.get("checksum") for r in records}) != 1:
        return "rccl", "direct results differ bit-wise between ranks"
    ref = max(r["ref_ms"] for r in records)
    direct = max(r["direct_ms"] for r in records)
    if not direct < ref:
        return "rccl", f"direct not faster ({direct:.3f} ms vs {ref:.3f} ms)"
    return "direct", f"direct correct and faster ({direct:.3f} ms vs {ref:.3f} ms)"


def probe_transports(process_group, device, make_direct, n_elems: int = 16 << 20, iters: int = 3,
                     fault_rank=None, timeout_s: float = 20.0):
    """SSL4GIE_ALLREDUCE=auto: all-reduce(mean) one test slice of `n_elems` floats through torch.distributed (RCCL on
    the GPU) and through the handle `make_direct()` returns (csrc/allreduce.hip), compare, time both (max over
    ranks), agree.  -> (direct handle or None, report dict).  Everything a rank can get wrong on its own — the
    construction of the handle, a peer that does not answer within `timeout_s` (the sticky-error path), a wrong
    sum — is caught, turned into direct_ok=False and exchanged through torch.distributed, so every rank reaches the
    same verdict and carries on in this process with the transport that works."""
    import time
    rank, world = dist.get_rank(process_group), dist.get_world_size(process_group)
    cuda = torch.device(device).type == "cuda"
    gen = torch.Generator("cpu").manual_seed(9000 + rank)
    x = torch.randn(min(n_elems, 1 << 22), generator=gen).repeat((n_elems + (1 << 22) - 1) // (1 << 22))[:n_elems].to(device)

    def timed(fn):
        """-> (ms per call, error or None); never raises and always reaches its barrier: the ranks' collective
        sequences stay aligned whatever one of them runs into"""
        err = None
        try:
            fn()  # warm-up (connections, first touch)
            if cuda:
                torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            err = e
        dist.barrier(group=process_group)
        t0 = time.perf_counter()
        if err is None:
            try:
                for _ in range(iters):
                    fn()
                if cuda:
                    torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001
                err = e
        return 1e3 * (time.perf_counter() - t0) / iters, err

    ref = x.clone()
    dist.all_reduce(ref, group=process_group)
    ref /= world
    buf = x.clone()
    ref_ms, _ = timed(lambda: dist.all_reduce(buf, group=process_group))
    rec = {"direct_ok": False, "ref_ms": ref_ms, "direct_ms": float("inf"), "checksum": None}
    direct = None
    try:
        direct = make_direct()
    except Exception as e:  # noqa: BLE001 — any failure means "not this transport"
        rec["error"] = f"init: {e}"

    def agree(flag):  # the construction and every later stage are collective: all ranks go on, or none
        flags = [None] * world
        dist.all_gather_object(flags, bool(flag), group=process_group)
        return all(flags)

    if agree(direct is not None):
        valid = False
        try:
            if hasattr(direct, "set_timeout"):
                direct.set_timeout(timeout_s)
            y = x.clone()
            direct.all_reduce_(y, 1.0 / world)
            if cuda:
                torch.cuda.synchronize()
            direct.raise_if_failed()
            if fault_rank is not None and rank == fault_rank:
                y[n_elems // 2] += 1.0
            err = float((y - ref).abs().max())
            scale = float(ref.abs().max())
            rec["max_abs_diff"] = err
            rec["checksum"] = (int(y.view(torch.int32).to(torch.int64).sum()), float(y.double().sum()))
            valid = err <= 1e-6 * max(scale, 1.0) and bool(torch.isfinite(y).all())
            if not valid:
                rec["error"] = f"max |direct - reference| = {err:.3e} (scale {scale:.3e})"
        except Exception as e:  # noqa: BLE001
            rec["error"] = f"run: {e}"
        if agree(valid):
            buf2 = x.clone()
            ms, terr = timed(lambda: (direct.all_reduce_(buf2, 1.0 / world), direct.raise_if_failed()))
            if terr is None:
                rec["direct_ms"], rec["direct_ok"] = ms, True
            else:
                rec["error"] = f"timing: {terr}"
        elif valid:
            rec["direct_ok"] = True   # this rank's result was right; a peer's was not (its record says so)
            rec["direct_ms"] = float("inf")
    else:
        rec.setdefault("error", "a peer could not create the handle")
    recs = [None] * world
    dist.all_gather_object(recs, rec, group=process_group)
    chosen, reason = choose_transport(recs)
    if chosen != "direct" and direct is not None:
        try:
            direct.close()
        except Exception:  # noqa: BLE001
            pass
        direct = None
    elif direct is not None and hasattr(direct, "set_timeout"):
        import os
        direct.set_timeout(float(os.environ.get("SSL4GIE_AR_TIMEOUT_S", "600")))
    fin = lambda v: None if v is None or v != v or v == float("inf") else round(v, 3)
    try:
        ref_backend = str(dist.get_backend(process_group))
    except Exception:  # noqa: BLE001
        ref_backend = "unknown"
    report = {"rccl_ms": fin(max(r["ref_ms"] for r in recs)),   # the torch.distributed transport (RCCL = "nccl"; gloo in rehearsals)
              "reference_backend": ref_backend,
              "direct_ms": fin(max(r["direct_ms"] for r in recs)),
              "mib": round(4 * n_elems / 2 ** 20, 1), "chosen": chosen, "reason": reason}
    return direct, report


class DirectAllReduce:
    """ssl4gie_allreduce_direct_* behind the interface DataParallel needs: one-hop reduce-scatter +
    all-gather over the node's xGMI links by peer-to-peer stores (csrc/allreduce.hip).  The IPC
    descriptions of the ranks' exchange regions travel through torch.distributed (any backend)."""

    def __init__(self, max_elems: int, process_group=None):
        import ctypes as C
        from . import _lib
        self.L = _lib.load()
        self._check = _lib.check
        self.rank = dist.get_rank(process_group)
        self.world = dist.get_world_size(process_group)
        nb = self.L.ssl4gie_allreduce_direct_blob_bytes()
        blob = C.create_string_buffer(nb)
        self.h = C.c_void_p()
        self._check(self.L.ssl4gie_allreduce_direct_init(self.rank, self.world, int(max_elems), blob, C.byref(self.h)),
                    "allreduce_direct_init")
        blobs = [None] * self.world
        dist.all_gather_object(blobs, bytes(blob.raw), group=process_group)
        self._check(self.L.ssl4gie_allreduce_direct_connect(self.h, b"".join(blobs)), "allreduce_direct_connect")
        dist.barrier(group=process_group)  # every rank has mapped every region before the first push
        self.max_elems = int(max_elems)

    def all_gather_(self, src: torch.Tensor, dst: torch.Tensor, stream=None):
        """dst [world * n] <- every rank's src [n] (fp32, contiguous), in rank order, enqueued on `stream`"""
        assert src.dtype == torch.float32 and dst.dtype == torch.float32 and src.is_cuda and dst.is_cuda
        assert src.is_contiguous() and dst.is_contiguous() and dst.numel() == self.world * src.numel()
        st = (stream or torch.cuda.current_stream()).cuda_stream
        self._check(self.L.ssl4gie_allgather_direct_enqueue(self.h, src.data_ptr(), src.numel(), dst.data_ptr(), st),
                    "allgather_direct_enqueue")

    def set_timeout(self, seconds: float):
        """bound of one in-kernel wait for a peer (default 600 s / SSL4GIE_AR_TIMEOUT_S)"""
        self._check(self.L.ssl4gie_allreduce_direct_set_timeout(self.h, float(seconds)), "allreduce_direct_set_timeout")

    def all_reduce_(self, t: torch.Tensor, scale: float, stream=None):
        """t (fp32, contiguous, 16-byte aligned) <- scale * sum over ranks, enqueued on `stream`"""
        assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda
        st = (stream or torch.cuda.current_stream()).cuda_stream
        n, off = t.numel(), 0
        while off < n:  # larger than the exchange region: several rounds
            m = min(self.max_elems, n - off)
            self._check(self.L.ssl4gie_allreduce_direct_enqueue(self.h, t.data_ptr() + 4 * off, m, float(scale), st),
                        "allreduce_direct_enqueue")
            off += m

    def raise_if_failed(self):
        """a peer that never signalled (poll time-out inside a kernel) leaves a sticky error word in
        the handle; enqueue returns it from then on and DataParallel raises instead of stepping on
        stale sums"""
        rc = self.L.ssl4gie_allreduce_direct_error(self.h) if self.h else 0
        if rc:
            raise RuntimeError(f"ssl4gie direct all-reduce: rank {(rc & 255) - 1} did not arrive in time in "
                               f"collective #{rc >> 8}; the gradients of that step were poisoned with NaN, "
                               "not reduced")

    def close(self):
        if self.h:
            torch.cuda.synchronize()
            self.L.ssl4gie_allreduce_direct_destroy(self.h)
            self.h = None


_SYNCBN_EXCHANGE = {}
SYNCBN_PROBE = {}   # id(process_group) -> report of the SSL4GIE_SYNCBN=auto probe


def probe_syncbn(process_group, device, make_direct, channels: int = 2048, iters: int = 20, timeout_s: float = 20.0):
    """SSL4GIE_SYNCBN=auto: one SyncBatchNorm record (2C + 1 floats) gathered through torch.distributed and through
    the direct exchange; the direct result must equal the reference bit for bit on every rank (an all-gather only
    moves data) and be faster.  Same never-raise, always-agree structure as probe_transports."""
    import time
    rank, world = dist.get_rank(process_group), dist.get_world_size(process_group)
    cuda = torch.device(device).type == "cuda"
    n = 2 * channels + 1
    src = torch.randn(n, generator=torch.Generator("cpu").manual_seed(7000 + rank)).to(device)
    ref = torch.empty(world * n, dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(ref, src, group=process_group)

    def timed(fn):
        err = None
        try:
            fn()
            if cuda:
                torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            err = e
        dist.barrier(group=process_group)
        t0 = time.perf_counter()
        if err is None:
            try:
                for _ in range(iters):
                    fn()
                if cuda:
                    torch.cuda.synchronize()
            except Exception as e:  # noqa: BLE001
                err = e
        return 1e3 * (time.perf_counter() - t0) / iters, err

    def agree(flag):
        flags = [None] * world
        dist.all_gather_object(flags, bool(flag), group=process_group)
        return all(flags)

    out = torch.empty_like(ref)
    ref_ms, _ = timed(lambda: dist.all_gather_into_tensor(out, src, group=process_group))
    rec = {"direct_ok": False, "ref_ms": ref_ms, "direct_ms": float("inf"), "checksum": None}
    direct = None
    try:
        direct = make_direct()
    except Exception as e:  # noqa: BLE001
        rec["error"] = f"init: {e}"
    if agree(direct is not None):
        valid = False
        try:
            if hasattr(direct, "set_timeout"):
                direct.set_timeout(timeout_s)
            got = torch.zeros_like(ref)
            direct.all_gather_(src, got)
            if cuda:
                torch.cuda.synchronize()
            direct.raise_if_failed()
            valid = bool(torch.equal(got, ref))
            rec["checksum"] = float(got.double().sum())
            if not valid:
                rec["error"] = "gathered records differ from torch.distributed's"
        except Exception as e:  # noqa: BLE001
            rec["error"] = f"run: {e}"
        if agree(valid):
            got2 = torch.empty_like(ref)
            ms, terr = timed(lambda: (direct.all_gather_(src, got2), direct.raise_if_failed()))
            if terr is None:
                rec["direct_ms"], rec["direct_ok"] = ms, True
            else:
                rec["error"] = f"timing: {terr}"
        elif valid:
            rec["direct_ok"] = True
    else:
        rec.setdefault("error", "a peer could not create the handle")
    recs = [None] * world
    dist.all_gather_object(recs, rec, group=process_group)
    chosen, reason = choose_transport(recs)
    if chosen != "direct" and direct is not None:
        try:
            direct.close()
        except Exception:  # noqa: BLE001
            pass
        direct = None
    elif direct is not None and hasattr(direct, "set_timeout"):
        import os
        direct.set_timeout(float(os.environ.get("SSL4GIE_AR_TIMEOUT_S", "600")))
    fin = lambda v: None if v is None or v != v or v == float("inf") else round(v, 4)
    return direct, {"torch_ms": fin(max(r["ref_ms"] for r in recs)), "direct_ms": fin(max(r["direct_ms"] for r in recs)),
                    "floats": n, "chosen": "direct" if chosen == "direct" else "torch.distributed", "reason": reason}


def syncbn_exchange(process_group=None, max_channels: int = 8192):
    """The node-local exchange SyncBatchNorm layers use when SSL4GIE_SYNCBN=direct (or =auto and the probe picked
    it): one DirectAllReduce handle of its own (the layers run on the compute stream; the gradient slices have
    theirs on the comm stream), created on first use — a collective, so every rank reaches it at the same layer.
    None when the option is off or the probe chose torch.distributed (which carries the statistics then)."""
    import os
    mode = os.environ.get("SSL4GIE_SYNCBN", "").lower()
    if mode not in ("direct", "auto") or not torch.cuda.is_available():
        return None
    key = id(process_group)
    if key not in _SYNCBN_EXCHANGE:
        world = dist.get_world_size(process_group)
        make = lambda: DirectAllReduce(world * (2 * max_channels + 4), process_group)
        if mode == "direct":
            _SYNCBN_EXCHANGE[key] = make()
        else:
            _SYNCBN_EXCHANGE[key], SYNCBN_PROBE[key] = probe_syncbn(process_group, torch.device("cuda", torch.cuda.current_device()), make)
    return _SYNCBN_EXCHANGE[key]


def comm_cus() -> int:
    """CUs left to the communication kernels in data-parallel runs (SSL4GIE_COMM_CUS, default 24: the persistent GEMM
    grids take whole rounds of tiles, and with the round-5 kernels the MAE step costs 22.24 ms at 232 and 240 compute
    CUs, 22.27 at 248, 22.5 at 208 / 216 but 22.8-23.0 at 224 — the former default of 32 sat on the one bad
    count: the decoder's fc1 / qkv products go from 7 / 5 rounds of tiles to 8 / 6; profiles/r05t_compute_cus_sweep.log)"""
    import os
    return max(0, min(128, int(os.environ.get("SSL4GIE_COMM_CUS", "24"))))


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
