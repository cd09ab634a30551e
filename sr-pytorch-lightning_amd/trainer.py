"""Minimal fit loop + data-parallel wiring for the `SRModel` surface (Lightning-optional).

The reference has no distributed code of its own: Lightning's Trainer picks DDP when `devices > 1`
(configs/all.yml:83,125-127) and torch DDP all-reduces the fp32 gradients over NCCL (SURVEY.md section 5,
8(e)).  Here: one process per GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI on ROCm) or
"gloo" on CPU, and a flat gradient buffer averaged in a few large buckets whose all-reduces overlap backward
(`GradSync`; xGMI is point-to-point: fewer, larger messages).  `wrap_ddp` (DistributedDataParallel, one bucket sized to the model, bucket views) is kept as the
`SRK_USE_TORCH_DDP=1` alternative.  The gradient mean is the only collective on the path.
"""
import os

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel as DDP


def dist_env():
    """(rank, world_size, local_rank) from the torchrun environment (1 process per GPU)."""
    return int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))


def init_distributed(device_type, force=False):
    """Create the process group when WORLD_SIZE > 1 (or `force`: a 1-rank group, used to exercise the DDP path on
    a single GPU)."""
    rank, world, local = dist_env()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if device_type == "cuda":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    return rank, world, local


def _flush_deferred():
    from . import ops
    ops.flush_wgrads()


def wrap_ddp(model, device, force=False):
    """DDP wrapper tuned for this path: one bucket (all grads), bucket views, no unused-parameter scan.
    torch's reducer copies each gradient into its bucket from inside backward, so the HIP path's deferred (grouped)
    weight gradients are switched off under it: every weight gradient is computed where autograd asks for it."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return model
    from . import ops
    prev = ops.set_defer_wgrad(False)    # process-wide while a DDP-wrapped model trains; `unwrap_ddp` restores what it was
    model.__dict__["_srk_prev_defer"] = prev
    nbytes = sum(p.numel() * 4 for p in model.parameters() if p.requires_grad)
    cap_mb = max(1, int(nbytes / 2 ** 20) + 1)
    kw = dict(gradient_as_bucket_view=True, bucket_cap_mb=cap_mb, broadcast_buffers=False, find_unused_parameters=False)
    if device.type == "cuda":
        return DDP(model, device_ids=[device.index], output_device=device.index, **kw)
    return DDP(model, **kw)


def unwrap_ddp(net):
    """Undo `wrap_ddp`'s process-wide switch once the wrapped model is done training."""
    from . import ops
    if isinstance(net, DDP):
        ops.set_defer_wgrad(net.module.__dict__.pop("_srk_prev_defer", True))      # (SRK_NO_DEFER_WGRAD=1 stays off)
        return net.module
    return net


class GradSync:
    """The path's only collective, without the DDP wrapper: the fp32 gradients live in ONE flat buffer cut into a few
    large buckets (reverse parameter order = the order backward produces them); a bucket's all-reduce (RCCL over xGMI
    on the GPUs, gloo on CPU) is launched asynchronously from a post-accumulate-grad hook as soon as its last gradient
    exists, so it OVERLAPS the rest of backward (what Lightning's DDP does for the reference, configs/all.yml:83,
    125-127), and `sync()` only waits.  Buckets are launched strictly in index order, so every rank issues the same
    sequence of collectives.  Same arithmetic as DistributedDataParallel (mean over ranks); what it drops is the
    reducer's per-step bookkeeping (0.65 ms of a 9 ms EDSR-baseline step on one rank).

    `overlap=False` registers no hooks: `sync()` then packs all gradients and issues the bucket all-reduces after
    backward -- the form a hipGraph-captured step uses (bench.py captures forward + backward + `pack()` in one graph,
    calls `reduce()` eagerly and then issues the optimizer step's three launches: no collective inside a capture).
    Replicas start identical: `broadcast()` sends rank 0's parameters and buffers."""

    def __init__(self, module, overlap=True, bucket_bytes=32 << 20, groups=None):
        """groups: optional list of parameter lists in the order their gradients become complete (the segments of a segmented
        backward pass, `OverlappedGraphStep`): buckets then never straddle two groups and `group_buckets[k]` lists group k's."""
        self.module = module
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.overlap = bool(overlap)
        if groups is None:
            order = list(reversed(self.params))          # backward reaches the last layers' parameters first
            group_of = {p: 0 for p in order}
        else:
            order = [p for g in groups for p in g]
            assert sorted(map(id, order)) == sorted(map(id, self.params)), "groups must partition the trainable parameters"
            group_of = {p: k for k, g in enumerate(groups) for p in g}
        self.flat = None
        self.views = {}
        self.buckets = []                                # [start, end, [params]]
        self.group_buckets = [[] for _ in range(1 if groups is None else len(groups))]
        total = sum(p.numel() for p in order)
        if order:
            self.flat = torch.zeros(total, dtype=torch.float32, device=order[0].device)
            off, cur, cur_g = 0, None, None
            for p in order:
                n = p.numel()
                if cur is None or group_of[p] != cur_g or (cur[1] - cur[0]) * 4 + n * 4 > bucket_bytes:
                    cur, cur_g = [off, off, []], group_of[p]
                    self.group_buckets[cur_g].append(len(self.buckets))
                    self.buckets.append(cur)
                self.views[p] = self.flat[off:off + n].view_as(p)
                cur[1] = off + n
                cur[2].append(p)
                off += n
        # the HIP weight-gradient kernels write a parameter's gradient straight into its slice (ops._grad_target): what is left
        # for `_pack_bucket` are the few gradients produced elsewhere (channel attention, PReLU, BatchNorm, torch ops)
        for p, v in self.views.items():
            p.__dict__["_srk_grad_target"] = v
        self._bucket_of = {p: i for i, b in enumerate(self.buckets) for p in b[2]}
        self._ready = [0] * len(self.buckets)
        self._next = 0                                   # first bucket not launched yet
        self._works = []
        self._avg = dist.is_initialized() and dist.get_backend() == "nccl"
        self._hooks = []
        if self.overlap and dist.is_initialized():
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
                p.__dict__["_srk_flush_aware"] = True     # this hook flushes ops' deferred weight gradients before it reads them

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self.params:
            p.__dict__.pop("_srk_flush_aware", None)

    def detach(self):
        """Undo everything this object attached to the parameters (hooks, gradient targets)."""
        self.remove_hooks()
        for p in self.params:
            p.__dict__.pop("_srk_grad_target", None)
            p.__dict__.pop("_srk_target_pass", None)

    def broadcast(self):
        if self.world > 1:
            with torch.no_grad():
                for t in list(self.module.parameters()) + list(self.module.buffers()):
                    dist.broadcast(t, src=0)

    # -- bucket plumbing ------------------------------------------------------------------------------------
    def _pack_bucket(self, i):
        """Gradients of bucket i -> their slices of the flat buffer (one multi-tensor copy); parameters without a
        gradient this step contribute zeros.  Afterwards p.grad IS the slice."""
        ps = self.buckets[i][2]
        src = [p.grad for p in ps if p.grad is not None and p.grad.data_ptr() != self.views[p].data_ptr()]
        dst = [self.views[p] for p in ps if p.grad is not None and p.grad.data_ptr() != self.views[p].data_ptr()]
        if src:
            torch._foreach_copy_(dst, src)
        for p in ps:
            if p.grad is None:
                self.views[p].zero_()
            else:
                p.grad = self.views[p]

    def _reduce_bucket(self, i, async_op):
        b = self.buckets[i]
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        return dist.all_reduce(self.flat[b[0]:b[1]], op=op, async_op=async_op)

    def _on_grad(self, p):
        i = self._bucket_of[p]
        self._ready[i] += 1
        while self._next < len(self.buckets) and self._ready[self._next] >= len(self.buckets[self._next][2]):
            _flush_deferred()            # the HIP path queues its weight gradients: compute them before they are read
            with torch.no_grad():
                self._pack_bucket(self._next)
                self._works.append(self._reduce_bucket(self._next, True))
            self._next += 1

    def pack(self):
        """All gradients -> the flat buffer (graph-capturable: no collective)."""
        with torch.no_grad():
            for i in range(len(self.buckets)):
                self._pack_bucket(i)

    def reduce(self):
        """Average the flat buffer over the ranks, bucket by bucket (after `pack()`)."""
        if not dist.is_initialized():
            return
        for i in range(len(self.buckets)):
            self._reduce_bucket(i, False)
        if not self._avg and self.world > 1:
            self.flat.mul_(1.0 / self.world)

    def sync(self):
        """Average the gradients over the ranks (call between backward and the optimizer step)."""
        if not dist.is_initialized() or self.flat is None:
            return
        _flush_deferred()
        with torch.no_grad():
            for i in range(self._next, len(self.buckets)):       # buckets the hooks did not complete (or no hooks)
                self._pack_bucket(i)
                self._works.append(self._reduce_bucket(i, True))
        for w in self._works:
            w.wait()
        self._works = []
        self._next = 0
        self._ready = [0] * len(self.buckets)
        if not self._avg and self.world > 1:
            self.flat.mul_(1.0 / self.world)


def auto_segments(model, bucket_bytes=32 << 20):
    """How many backward segments the graph-replayed multi-rank step uses: SRK_DDP_SEGMENTS if set, else one per ~32 MB of
    gradients (at most 8) once there are more than 48 MB of them -- below that one all-reduce is latency-bound and extra
    graph launches (~15 us each) cost more than the overlap returns."""
    if not getattr(model, "supports_backward_segments", True):       # (WDSR: one autograd node owns every weight norm)
        return 1
    env = os.environ.get("SRK_DDP_SEGMENTS", "auto")
    if env != "auto":
        return max(1, int(env))
    nbytes = sum(p.numel() * 4 for p in model.parameters() if p.requires_grad)
    return 1 if nbytes <= (48 << 20) else min(8, -(-nbytes // bucket_bytes))


def synthetic_batch(n, channels, lr_size, scale, seed, device):
    """The BASELINE workload: uniform [0,1) LR patches and HR targets."""
    g = torch.Generator().manual_seed(seed)
    lr = torch.rand(n, channels, lr_size, lr_size, generator=g)
    hr = torch.rand(n, channels, lr_size * scale, lr_size * scale, generator=g)
    return {"lr": lr.to(device), "hr": hr.to(device), "path": [f"synthetic/{i}" for i in range(n)]}


def _eager_fwd_bwd_reduce(model, net, optimizer, gsync, scaler, batch):
    """An eager step WITHOUT its optimizer step: forward, losses, backward, gradient average (what primes `GraphedStep`'s
    optimizer-first graph)."""
    from . import ops
    ops.discard_wgrads()
    optimizer.zero_grad(set_to_none=True)
    loss = model._calculate_losses(img_sr=net(batch["lr"]), img_hr=batch["hr"])["loss"]
    ops.backward(loss if (scaler is None or not hasattr(scaler, "state")) else scaler.scale(loss))
    ops.flush_wgrads()
    if gsync is not None:
        gsync.sync()
    return loss.detach()


def _eager_step(model, net, optimizer, gsync, scaler, batch):
    """forward -> losses -> backward -> [gradient average] -> optimizer step, launch by launch."""
    from . import ops
    ops.discard_wgrads()                 # nothing of an earlier pass that ended abnormally may linger in the deferred queue
    optimizer.zero_grad(set_to_none=True)
    img_sr = net(batch["lr"])
    loss = model._calculate_losses(img_sr=img_sr, img_hr=batch["hr"])["loss"]
    if scaler is not None and hasattr(scaler, "state"):      # optim.DeviceGradScaler: check / skip / unscale inside the optimizer's launch
        ops.backward(scaler.scale(loss))
        ops.flush_wgrads()
        if gsync is not None:
            gsync.sync()
        optimizer.step(grad_scaler=scaler)
    elif scaler is not None:                                  # torch.amp.GradScaler (an optimizer that is not the HIP Adam)
        scaler.scale(loss).backward()
        ops.flush_wgrads()
        if gsync is not None:
            gsync.sync()
        scaler.step(optimizer)
        scaler.update()
    else:
        ops.backward(loss)
        ops.flush_wgrads()               # normally a no-op: the engine's final callback already ran
        if gsync is not None:
            gsync.sync()
        optimizer.step()
    # detached: an autograd graph of an eager step that is still referenced when a later step is CAPTURED gets released inside
    # the capture (when its holder is reassigned), which crashes hipStreamEndCapture on this stack
    return loss.detach()


class GraphedStep:
    """The training step as hipGraph replays.

    At the reference's batch (16 patches of 48x48 per GPU) a step is hundreds of launches of 5-20 us; issued one by one from
    Python the host is the bottleneck (RCAN: 60 ms per step eager, 10 ms replayed).  The first `warm_steps` batches run eagerly
    (they are ordinary training steps: nothing is repeated or skipped); the next batch is copied into static buffers and the
    step is CAPTURED (capturing records launches, it does not execute them) and then replayed for it and every later batch of
    the same shape.  One process: one graph (forward, loss, backward, Adam).  Several ranks: forward + backward + gradient
    packing are one graph, the bucketed all-reduce (RCCL) and the optimizer step (three launches) are issued eagerly behind it --
    no collective is ever inside a capture.  Round 4's form of it: ONE graph = [the update of the PREVIOUS step] + forward + backward +
    packing, so between two calls the parameters LAG one update; `flush()` applies it (call it before anything reads the parameters:
    validation, checkpoints; `finish()` at the end of training), always with the hyper-parameters its gradients were produced under.
    A batch of another shape (a short last batch) runs eagerly.  If a capture fails
    the loop stays eager (and says so once).  Callers must not keep a loss WITH its autograd graph from an earlier eager step
    alive across the capture (`_eager_step` returns it detached for that reason)."""

    def __init__(self, model, net, optimizer, gsync, warm_steps=3, scaler=None):
        self.model, self.net, self.opt, self.gsync = model, net, optimizer, gsync
        self.scaler = scaler                 # optim.DeviceGradScaler (fp16) or None: its launches are part of the captured step
        self.pending = False                 # several ranks, graph form: reduced gradients whose optimizer step opens the NEXT replay
        self.warm_steps = int(warm_steps)
        self.segments = auto_segments(model) if gsync is not None else 1
        self.ogs = None                      # OverlappedGraphStep (several ranks, large models): all-reduces beside backward
        self.seen = 0
        self.graphs = None
        self.static = None
        self.loss = None
        self.failed = False
        self.hyper = None
        self.pending_hyper = None            # the hyper-parameters that were live when the pending gradients were produced

    def _hyper(self):
        """The optimizer's hyper-parameters travel BY VALUE in the captured launch (srk_adam_args): a scheduler or a manual
        change after the capture would be ignored by a replay, so a change re-captures."""
        keys = ("lr", "betas", "eps", "weight_decay", "maximize", "momentum", "alpha")
        return tuple(tuple((k, tuple(g[k]) if isinstance(g[k], (tuple, list)) else float(g[k])) for k in keys if k in g)
                     for g in self.opt.param_groups)

    def _set_hyper(self, snap):
        for g, vals in zip(self.opt.param_groups, snap):
            for k, v in vals:
                g[k] = tuple(v) if isinstance(v, tuple) else v

    def _mark_pending(self):
        self.pending, self.pending_hyper = True, self._hyper()

    def flush(self):
        """Apply the update that is still pending (several ranks, optimizer-first graph form: step k's update normally opens replay
        k + 1, so between two calls the parameters LAG one update).  Call before reading the parameters -- validation, a checkpoint,
        `state_dict()` -- and when training ends (`finish()` = this).  The update uses the hyper-parameters that were live when its
        gradients were produced: the reference's order is `optimizer.step()`, then `scheduler.step()`, so a learning rate changed
        after step k must not scale step k's update."""
        if not self.pending:
            return
        cur = self._hyper()
        if self.pending_hyper is not None and self.pending_hyper != cur:
            self._set_hyper(self.pending_hyper)
            try:
                self._opt_step()
            finally:
                self._set_hyper(cur)
        else:
            self._opt_step()
        self.pending, self.pending_hyper = False, None

    def shifted_eager_step(self, batch):
        """What ONE replay of the optimizer-first graph does, launch by launch: [the update the previous step left pending] forward,
        losses, backward, gradient average -- and this step's update stays pending.  Used for batches the graph does not fit (a short
        last batch) and by tests/test_ddp_gloo.py, which drives the shifted order on two CPU ranks."""
        self.flush()
        loss = _eager_fwd_bwd_reduce(self.model, self.net, self.opt, self.gsync, self.scaler, batch)
        self._mark_pending()
        return loss

    def _fwd_bwd(self):
        self.opt.zero_grad(set_to_none=True)
        sr = self.net(self.static["lr"])
        loss = self.model._calculate_losses(img_sr=sr, img_hr=self.static["hr"])["loss"]
        from . import ops
        ops.backward(loss if self.scaler is None else self.scaler.scale(loss))
        return loss

    def _opt_step(self):
        if self.scaler is None:
            self.opt.step()
        else:
            self.opt.step(grad_scaler=self.scaler)

    def _capture(self, batch):
        from . import ops
        self.static = {"lr": batch["lr"].clone(), "hr": batch["hr"].clone()}
        self.hyper = self._hyper()
        if hasattr(self.opt, "reserve_capture_tables"):
            self.opt.reserve_capture_tables()    # page-locked staging buffers cannot be allocated inside the capture
        if self.gsync is not None:
            self.gsync.remove_hooks()            # no collective from inside backward any more: pack() / reduce() around the graphs
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        cem = "thread_local" if self.gsync is not None else "global"      # RCCL's watchdog polls events while this thread captures
        if self.gsync is None:
            g = torch.cuda.CUDAGraph()
            with ops.graph_capture(g, stream=side, capture_error_mode=cem):
                self.loss = self._fwd_bwd()
                self._opt_step()
            self.graphs = (g,)
        else:
            # ONE graph per step: [optimizer step on the gradients the previous step reduced] + forward + backward + packing; only the
            # bucket all-reduces are issued eagerly between two replays (round 3 also issued the optimizer step's three launches
            # eagerly: host-paced, half of the structure's overhead at batch 16).  Same arithmetic, shifted by one replay: the
            # caller's batch trains eagerly up to the reduce (its update opens the first replay), `finish()` applies the last one.
            self.flush()                     # (a re-capture after a hyper-parameter change: the previous replay's update first, with ITS values)
            self.primed = None
            now = _eager_fwd_bwd_reduce(self.model, self.net, self.opt, self.gsync, self.scaler, self.static)
            self._mark_pending()
            self.primed = now                # this batch has trained up to its reduce: a capture failure below must not train it again
            if hasattr(self.opt, "reserve_capture_tables"):
                self.opt.reserve_capture_tables()
            torch.cuda.synchronize()
            side.wait_stream(torch.cuda.current_stream())
            ga = torch.cuda.CUDAGraph()
            with ops.graph_capture(ga, stream=side, capture_error_mode=cem):
                self._opt_step()
                self.loss = self._fwd_bwd()
                self.gsync.pack()
            self.graphs = (ga, "opt_first")
            self.primed = None
            return now
        return None

    def finish(self):
        """Call when training ends: the update of the last replayed step (multi-rank graph form) is still pending."""
        self.flush()

    def __call__(self, batch):
        self.seen += 1
        same = self.static is not None and batch["lr"].shape == self.static["lr"].shape and batch["hr"].shape == self.static["hr"].shape
        if self.failed or self.seen <= self.warm_steps or (self.graphs is not None and not same):
            if self.ogs is not None and self.ogs.gsync is not None:
                return self.ogs.eager_step(batch)          # (the buckets were re-cut along the segments: its own eager form)
            if self.pending:                               # optimizer-first graph form: keep "reduced gradients, update pending" invariant
                return self.shifted_eager_step(batch)
            return _eager_step(self.model, self.net, self.opt, self.gsync, self.scaler, batch)
        if self.graphs is not None and self._hyper() != self.hyper:
            # lr / betas / ... changed: capture again with the new values (the old graphs go first, then the tables they read).  The
            # update still pending belongs to the OLD values (`flush`, called by `_capture`)
            self.graphs = None
            if self.ogs is not None:
                self.ogs.graphs = None
            if hasattr(self.opt, "release_captured_tables"):
                torch.cuda.synchronize()
                self.opt.release_captured_tables()
        if self.graphs is None and self.segments > 1 and self.gsync is not None:
            if self.ogs is None:                 # (two ordinary eager steps: the segments' parameter groups, the re-cut buckets)
                bucket_bytes = int(os.environ.get("SRK_BUCKET_BYTES", 32 << 20))
                try:
                    self.gsync.detach()
                    ogs = OverlappedGraphStep(self.model, self.opt, self.segments, bucket_bytes=bucket_bytes, scaler=self.scaler)
                    loss = ogs.prepare(batch)
                    self.ogs = ogs
                    return loss
                except SegmentationUnavailable as e:     # every rank raises it together (prepare() agrees on the outcome before any collective)
                    import sys
                    from . import ops
                    ops.discard_wgrads()
                    print(f"[trainer] segmented backward not possible for this model ({type(e).__name__}: {e}); one backward graph", file=sys.stderr)
                    self.segments, self.ogs = 1, None
                    self.gsync = GradSync(self.model, bucket_bytes=bucket_bytes)
                    return _eager_step(self.model, self.net, self.opt, self.gsync, self.scaler, batch)
            try:
                torch.cuda.synchronize()
                self.ogs.capture(batch)
                self.static, self.graphs, self.hyper = self.ogs.static, tuple(self.ogs.graphs), self._hyper()
                return self.ogs.step()
            except Exception as e:  # noqa: BLE001
                import sys
                from . import ops
                ops.discard_wgrads()
                print(f"[trainer] segmented hipGraph capture failed ({type(e).__name__}: {e}); training continues eagerly", file=sys.stderr)
                self.failed, self.graphs, self.static = True, None, None
                torch.cuda.synchronize()
                return self.ogs.eager_step(batch)
        if self.ogs is not None and self.graphs is not None:
            return self.ogs.step(batch)
        if self.graphs is None:
            try:
                torch.cuda.synchronize()
                now = self._capture(batch)
            except Exception as e:  # noqa: BLE001
                import sys
                from . import ops
                ops.discard_wgrads()
                print(f"[trainer] hipGraph capture failed ({type(e).__name__}: {e}); training continues eagerly", file=sys.stderr)
                self.failed, self.graphs, self.static = True, None, None
                torch.cuda.synchronize()
                primed, self.primed = getattr(self, "primed", None), None
                self.flush()
                if primed is not None:       # the batch already trained eagerly up to its reduce inside _capture: its update was the flush
                    return primed
                return _eager_step(self.model, self.net, self.opt, self.gsync, self.scaler, batch)
            if now is not None:              # several ranks: this batch trained eagerly up to the reduce; its update opens the first replay
                return now
        else:
            self.static["lr"].copy_(batch["lr"], non_blocking=True)
            self.static["hr"].copy_(batch["hr"], non_blocking=True)
        self.graphs[0].replay()
        if len(self.graphs) == 2:            # [pending update] forward, backward, packing were the graph; the collective is eager
            self.gsync.reduce()
            self.pending_hyper = self.hyper  # (the replay applied the previous update with the captured values and left a new one pending)
        return self.loss.detach()


class SegmentationUnavailable(RuntimeError):
    """Raised by `OverlappedGraphStep.prepare` on EVERY rank when the segmented backward pass failed on any of them (the ranks
    agree through one MIN all-reduce before any gradient collective), so that all of them take the same fallback."""


class OverlappedGraphStep:
    """The multi-rank training step as hipGraph replays WITH the gradient all-reduces overlapping backward.

    `GraphedStep`'s multi-rank form replays forward + backward as one graph and only then starts the bucket all-reduces: the
    communication is exposed (for EDSR-large, 172 MB of gradients, ~1 ms of a 13 ms batch-16 step on 8 GPUs).  A capture cannot be
    ended and another begun from inside `loss.backward()`, so the cut is made in the autograd graph instead: the models mark block
    boundaries with `ops.cut`, `ops.record_segments(every)` detaches there, and `ops.backward_segments` runs the backward pass as
    K + 1 independent autograd passes.  Captured: ONE graph for forward + loss, one graph per backward segment (a shared memory
    pool; each also packs the few gradients that were not written into the flat buffer directly).  Replayed: forward graph, then
    per segment its graph followed by the ASYNCHRONOUS all-reduces of that segment's buckets (RCCL orders itself behind the
    segment on the launch stream and runs beside the next segment's graph), then one wait and the optimizer step's three
    launches.  The buckets are re-cut along the segments (`GradSync(groups=...)`), every rank the same way.

    prepare(batch) is ONE ordinary eager training step (it finds which parameters each segment completes and re-cuts the buckets)."""

    def __init__(self, model, optimizer, segments, bucket_bytes=32 << 20, scaler=None):
        self.model, self.opt, self.want = model, optimizer, max(2, int(segments))
        self.scaler = scaler                 # optim.DeviceGradScaler or None
        self.bucket_bytes = bucket_bytes
        self.gsync = self.graphs = self.static = self.loss = None
        self.every, self.limit = 1, None

    def _forward(self, batch, every):
        from . import ops
        with ops.record_segments(every, limit=self.limit) as rec:
            loss = self.model._calculate_losses(img_sr=self.model(batch["lr"]), img_hr=batch["hr"])["loss"]
            if self.scaler is not None:      # the backward passes start from the scaled loss; callers report `self.raw_loss`
                self.raw_loss = loss.detach()
                loss = self.scaler.scale(loss)
            else:
                self.raw_loss = loss.detach()
        return loss, rec

    def _opt_step(self):
        if self.scaler is None:
            self.opt.step()
        else:
            self.opt.step(grad_scaler=self.scaler)

    def prepare(self, batch):
        """Counts the model's cut points, picks the stride, finds the parameter groups with one eager training step on `batch`
        and builds the GradSync along them."""
        from . import ops
        # the probe forward only counts cut points: BatchNorm running statistics / num_batches_tracked must not see the batch twice
        bufs = [(b, b.detach().clone()) for b in self.model.buffers()]
        with torch.no_grad():
            _, rec = self._forward(batch, 1)
            for b, keep in bufs:
                b.copy_(keep)
        ncut = rec.count
        # the model's mandatory cuts (long skip connections, `ops.cut(..., keep=True)`) count towards the wanted segments; the
        # optional ones (block boundaries) are thinned to spread the rest evenly
        self.limit = max(0, self.want - 1 - rec.keeps)
        self.every = max(1, ncut // (self.limit + 1)) if ncut else 1
        params = [p for p in self.model.parameters() if p.requires_grad]
        first, sums = {}, []                                  # id -> segment of the first gradient; per segment: checksums of every gradient so far

        def collect(k):
            ops.flush_wgrads()
            for p in params:
                if p.grad is not None and id(p) not in first:
                    first[id(p)] = k
            have = [p for p in params if p.grad is not None]
            sums.append((have, torch.stack([p.grad.detach().double().sum() for p in have]) if have else None))
        # The local part may fail on ONE rank only (out of memory, a HIP error): the ranks agree on the outcome BEFORE the gradient
        # collective below -- a rank that fell back alone would issue bucketed all-reduces against the others' single flat one
        # (mismatched collectives hang or corrupt the gradients).
        err = None
        try:
            self.opt.zero_grad(set_to_none=True)
            loss, rec = self._forward(batch, self.every)
            ops.backward_segments(loss, rec.cuts, after=collect)
        except Exception as e:                 # autograd's "backward through the graph a second time", HIP / allocation errors -- and ANY other
            err = e                            # failure of a model or hook: a rank that left here would leave the others in the all-reduce below
            ops.discard_wgrads()
        if dist.is_initialized() and dist.get_world_size() > 1:
            flag = torch.tensor([0.0 if err is not None else 1.0], device=next(self.model.parameters()).device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if float(flag.item()) == 0.0:
                raise SegmentationUnavailable(str(err) if err is not None else "the segmented backward pass failed on another rank")
        elif err is not None:
            raise SegmentationUnavailable(str(err))
        # a parameter used on both sides of a cut (a shared module) keeps changing after its first segment: its bucket may only be
        # reduced behind the LAST segment that touches it (one host read for all checksums)
        last = dict(first)
        prev = {}
        for k, (have, cs) in enumerate(sums):
            vals = cs.tolist() if cs is not None else []
            for p, v in zip(have, vals):
                if id(p) in prev and prev[id(p)] != v:
                    last[id(p)] = k
                prev[id(p)] = v
        nseg = len(sums)
        groups = [[p for p in params if last.get(id(p)) == k] for k in range(nseg)]
        groups.append([p for p in params if id(p) not in last])   # parameters without a gradient this step: one more group at the end
        if dist.is_initialized() and dist.get_world_size() > 1:   # the first step's gradients still have to be averaged: ONE collective
            have = [p for p in params if p.grad is not None]
            if have:
                flat = torch.cat([p.grad.reshape(-1) for p in have])
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
                flat.div_(dist.get_world_size())
                off = 0
                for p in have:
                    n = p.numel()
                    p.grad.copy_(flat[off:off + n].view_as(p.grad))
                    off += n
        self._opt_step()
        self.groups = groups
        self.gsync = GradSync(self.model, overlap=False, bucket_bytes=self.bucket_bytes, groups=groups)
        return self.raw_loss

    def _reduce_group(self, k, works):
        for i in self.gsync.group_buckets[k] if k < len(self.gsync.group_buckets) else ():
            works.append(self.gsync._reduce_bucket(i, True))

    def _finish(self, works):
        for w in works:
            w.wait()
        if not self.gsync._avg and self.gsync.world > 1:
            self.gsync.flat.mul_(1.0 / self.gsync.world)
        self._opt_step()

    def eager_step(self, batch):
        """The same step launch by launch (warm-up, batches of another shape)."""
        from . import ops
        self.opt.zero_grad(set_to_none=True)
        loss, rec = self._forward(batch, self.every)
        works = []

        def after(k):
            ops.flush_wgrads()
            with torch.no_grad():
                for i in (self.gsync.group_buckets[k] if k < len(self.gsync.group_buckets) else ()):
                    self.gsync._pack_bucket(i)
            self._reduce_group(k, works)
        ops.backward_segments(loss, rec.cuts, after=after)
        with torch.no_grad():
            for k in range(len(rec.cuts) + 1, len(self.gsync.group_buckets)):      # the group of gradient-less parameters
                for i in self.gsync.group_buckets[k]:
                    self.gsync._pack_bucket(i)
                self._reduce_group(k, works)
        self._finish(works)
        return self.raw_loss

    def capture(self, batch, capture_error_mode="thread_local"):
        from . import ops
        self.static = {"lr": batch["lr"].clone(), "hr": batch["hr"].clone()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        self.opt.zero_grad(set_to_none=True)
        gf = torch.cuda.CUDAGraph()
        with ops.graph_capture(gf, stream=side, capture_error_mode=capture_error_mode):
            loss, rec = self._forward(self.static, self.every)
        pool = gf.pool()
        graphs = [gf]
        cuts = rec.cuts
        nseg = len(cuts) + 1
        for k in range(nseg):
            g = torch.cuda.CUDAGraph()
            with ops.graph_capture(g, stream=side, pool=pool, capture_error_mode=capture_error_mode):
                if k == 0:
                    loss.backward()
                else:
                    pairs = cuts[nseg - 1 - k]
                    roots = [o for o, l in pairs if l.grad is not None]
                    if roots:
                        torch.autograd.backward(roots, [l.grad for o, l in pairs if l.grad is not None])
                ops.flush_wgrads()
                with torch.no_grad():
                    ks = [k] if k < nseg - 1 else list(range(k, len(self.gsync.group_buckets)))
                    for kk in ks:
                        for i in self.gsync.group_buckets[kk]:
                            self.gsync._pack_bucket(i)
            graphs.append(g)
        self.graphs, self.loss, self.nseg = graphs, self.raw_loss, nseg
        self._keep = (rec, cuts)                         # the cut tensors are the graphs' static memory

    def step(self, batch=None):
        if batch is not None:
            self.static["lr"].copy_(batch["lr"], non_blocking=True)
            self.static["hr"].copy_(batch["hr"], non_blocking=True)
        self.graphs[0].replay()
        works = []
        for k in range(self.nseg):
            self.graphs[1 + k].replay()
            ks = [k] if k < self.nseg - 1 else range(k, len(self.gsync.group_buckets))
            for kk in ks:
                self._reduce_group(kk, works)
        self._finish(works)
        return self.loss.detach()


class Trainer:
    """`fit(model, batches)`: forward -> losses -> backward -> optimizer.step, DDP when WORLD_SIZE > 1.

    Mirrors what Lightning's fit loop does around `SRModel.training_step` (srmodel.py:160-171):
    nothing else (no checkpointing / loggers -- out of scope)."""

    def __init__(self, device=None, max_steps=-1, log_every=0, use_grad_scaler=None, use_graph=None):
        if device is None:
            device = torch.device("cuda", dist_env()[2]) if torch.cuda.is_available() else torch.device("cpu")
        self.device = torch.device(device)
        self.max_steps = max_steps
        self.log_every = log_every
        self.use_grad_scaler = use_grad_scaler
        # hipGraph replay of the training step (GraphedStep): on by default on a GPU; SRK_TRAIN_GRAPH=0 or use_graph=False: eager
        self.use_graph = (os.environ.get("SRK_TRAIN_GRAPH", "1") != "0") if use_graph is None else bool(use_graph)
        self.rank, self.world, self.local = init_distributed(self.device.type)
        self._loss_host = []             # floats already fetched
        self._loss_dev = []              # device scalars of the latest steps (no host sync per step: replays queue ahead)

    @property
    def losses(self):
        """Per-step loss values.  Reading them synchronises with the device (once per read, not once per step)."""
        if self._loss_dev:
            self._loss_host += torch.stack(self._loss_dev).float().cpu().tolist()
            self._loss_dev = []
        return self._loss_host

    def fit(self, model, batches):
        model = model.to(self.device)
        use_ddp = os.environ.get("SRK_USE_TORCH_DDP") == "1"
        net = wrap_ddp(model, self.device) if use_ddp else model
        gsync = None
        if not use_ddp and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            gsync = GradSync(model, bucket_bytes=int(os.environ.get("SRK_BUCKET_BYTES", 32 << 20)))
            gsync.broadcast()
        optimizer = model.configure_optimizers()[0]
        scaler = None
        use_scaler = self.use_grad_scaler
        if use_scaler is None:
            use_scaler = getattr(model, "compute_dtype", torch.float32) == torch.float16 and self.device.type == "cuda"
        if use_scaler:
            # fp16 (the reference's `precision: 16` = Lightning "16-mixed": autocast + GradScaler, configs/all.yml:122): dynamic loss
            # scaling with its state on the device when the optimizer is the HIP Adam, so that the step can still be a hipGraph
            from .optim import Adam as _HipAdam, DeviceGradScaler
            scaler = DeviceGradScaler(self.device) if isinstance(optimizer, _HipAdam) else torch.amp.GradScaler("cuda")
        self.scaler = scaler
        graphed = None
        if self.use_graph and (scaler is None or hasattr(scaler, "state")) and self.device.type == "cuda" and not use_ddp:
            graphed = GraphedStep(model, net, optimizer, gsync, warm_steps=11 if gsync is not None else 3, scaler=scaler)
        try:
            for step, batch in enumerate(batches):
                if 0 <= self.max_steps <= step:
                    break
                batch = {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
                if graphed is not None:
                    loss = graphed(batch)
                else:
                    loss = _eager_step(model, net, optimizer, gsync, scaler, batch)
                # a replayed step returns the graph's static loss tensor: clone (one tiny asynchronous launch), never float()
                self._loss_dev.append(loss.detach().clone())
                if self.log_every and (step + 1) % self.log_every == 0:
                    last = self.losses[-1]
                    if self.rank == 0:
                        print(f"step {step + 1}: loss {last:.6f}", flush=True)
                elif len(self._loss_dev) >= 1024:
                    self.losses  # noqa: B018  (drain to the host list)
            if graphed is not None:
                graphed.finish()         # (multi-rank graph form: the last replay's update)
        finally:                 # whatever ended the loop: nothing stays attached to the parameters, the deferral switch is what it was
            self.graphed = graphed
            for gs in {id(g): g for g in (gsync, getattr(graphed, "gsync", None), getattr(getattr(graphed, "ogs", None), "gsync", None)) if g is not None}.values():
                gs.detach()
            unwrap_ddp(net)
        return model
