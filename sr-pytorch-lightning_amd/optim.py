"""Optimizers of the training hot path on the HIP kernels.

`Adam` is torch.optim.Adam -- the optimizer the reference builds (models/srmodel.py:57-64 `_supported_optimizers['ADAM']`,
:145-154 `configure_optimizers`, always at torch's defaults because :602-603 drops the user's parameters) -- with the whole
update as ONE launch of `srk_adam_step` (csrc/optim.hip) instead of one launch per few dozen tensors.  Same constructor,
same `state_dict()` layout (`step`, `exp_avg`, `exp_avg_sq` per parameter), same arithmetic in fp32.  The step count lives
on the device, so `step()` can be captured into a hipGraph and replayed.

Nothing here falls back to PyTorch arithmetic: parameters must be fp32 tensors on a GPU and the HIP library must load.
"""
import ctypes as C

import torch

from . import _lib as L

_CHUNK = 4096           # elements per workgroup (256 threads x 4 x float4)


class DeviceGradScaler:
    """Dynamic loss scaling whose state lives on the GPU: torch.amp.GradScaler's arithmetic (init_scale 2^16, growth factor 2,
    backoff 0.5, growth interval 2000 -- what Lightning's "16-mixed" precision wraps around the reference's training_step for
    `precision: 16`, configs/all.yml:122) without its per-step host read.  GradScaler.step() asks the HOST whether a gradient was
    non-finite and then calls or skips optimizer.step() from Python; a hipGraph replay cannot branch on the host, so fp16 training
    ran launch by launch (RCAN: 72.8 ms per step against 19 ms replayed).  Here the check, the skip and the scale update are
    launches of the step itself (`srk_adam_step_scaled`, `srk_loss_scale_update`):

        loss_s = scaler.scale(loss)            # loss * scale (a device scalar: the backward pass carries it)
        loss_s.backward()
        optimizer.step(grad_scaler=scaler)     # non-finite gradient anywhere -> nothing is updated; scale halves; else g / scale

    state (8 device floats): scale, growth tracker, found_inf, growth factor, backoff factor, growth interval, skipped steps, -."""

    def __init__(self, device, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.state = torch.tensor([init_scale, 0.0, 0.0, growth_factor, backoff_factor, float(growth_interval), 0.0, 0.0],
                                  dtype=torch.float32, device=device)

    def scale(self, loss):
        return loss * self.state[0]            # a view of the state: no host read, capturable

    def get_scale(self):
        return float(self.state[0])            # (host read: logging only)

    @property
    def skipped_steps(self):
        return int(self.state[6])

    def state_dict(self):
        return {"scale": self.get_scale(), "growth_tracker": int(self.state[1])}

    def load_state_dict(self, d):
        self.state[0] = float(d["scale"])
        self.state[1] = float(d.get("growth_tracker", 0))


class _Table:
    """One device copy of the (parameter, gradient) table with its page-locked staging buffer.  The eager steps own one;
    every hipGraph capture gets ITS OWN (a captured upload is a memcpy node that re-reads the staging buffer on every
    replay, so nothing an eager step does later may touch the bytes a live graph points at: ADVICE r2)."""

    def __init__(self, cap, dev):
        self.table = torch.empty(cap, dtype=torch.uint8, device=dev)   # device bytes: slots | blocks
        self.host = torch.empty(cap, dtype=torch.uint8, pin_memory=True)
        self.copied = None      # event behind the last eager upload from `host`
        self.key = None
        self.grads = None       # the gradient tensors the table was built for (kept alive: their addresses are in it)
        self.live = []          # the parameters in the table
        self.live_ids = None    # their identities (same set, moved addresses: only the pointer fields are rewritten)
        self.slots = None
        self.nslots = self.nblocks = self.blocks_off = 0


class _Plan:
    """Per parameter group: flat moment buffers, the device step counter and the device tables of (parameter, gradient)."""

    def __init__(self):
        self.offsets = {}       # param -> offset (floats) into the flat moment buffers
        self.index = {}         # param -> index into `steps`
        self.m = self.v = None
        self.steps = self.ticket = None  # per-parameter update counts (torch counts per parameter)
        self.cap = 0
        self.eager = None       # _Table of the launch-by-launch steps
        self.captured = []      # _Tables owned by hipGraph captures: written once, kept alive as long as the plan
        self.spare = None       # allocated OUTSIDE any capture (page-locked allocations are not capturable) for the next one


class Adam(torch.optim.Adam):
    """torch.optim.Adam(params, lr, betas, eps, weight_decay, amsgrad) with a single-launch step for GPU parameters.

    `amsgrad=True` is not implemented (the reference never sets it); `maximize` and L2 `weight_decay` are.  A model whose
    parameters live on the CPU (SRCNN, the reference's CPU-runnable case, runs on torch ops there) steps through
    torch.optim.Adam itself; GPU parameters always take the HIP kernel and raise if the library is missing."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, *, maximize=False):
        if not 0.0 <= lr:
            raise ValueError(f"Invalid learning rate: {lr}")
        if not 0.0 <= eps:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 0: {betas[0]}")
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 1: {betas[1]}")
        if not 0.0 <= weight_decay:
            raise ValueError(f"Invalid weight_decay value: {weight_decay}")
        if amsgrad:
            raise NotImplementedError("amsgrad is not implemented by the HIP Adam (the reference never enables it)")
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=maximize)
        self._plans = {}

    # -- state ------------------------------------------------------------------------------------
    def _plan(self, gi, group):
        plan = self._plans.get(gi)
        params = [p for p in group["params"] if p.requires_grad]
        if plan is not None and all(p in plan.offsets for p in params):
            return plan
        if not params:
            return None
        dev = params[0].device
        for p in params:
            if not p.is_cuda:
                raise RuntimeError("sr-pytorch-lightning_amd.optim.Adam runs on the GPU only (parameters are on %s); there is no CPU path" % p.device)
            if p.dtype != torch.float32 or not p.is_contiguous() or p.device != dev:
                raise RuntimeError("HIP Adam needs contiguous fp32 parameters on one device")
        plan = _Plan()
        off = 0
        for k, p in enumerate(params):
            plan.offsets[p], plan.index[p] = off, k
            off += (p.numel() + 3) // 4 * 4                 # every tensor starts 16-byte aligned
        plan.m = torch.zeros(max(off, 4), dtype=torch.float32, device=dev)
        plan.v = torch.zeros(max(off, 4), dtype=torch.float32, device=dev)
        plan.steps = torch.zeros(len(params), dtype=torch.float32, device=dev)
        plan.ticket = torch.zeros(1, dtype=torch.int32, device=dev)
        # table of the largest case (every parameter has a gradient): tensors | 16-byte gap | blocks
        cap = C.sizeof(L.AdamSlot) * len(params) + 16 + C.sizeof(L.AdamBlock) * sum((p.numel() + _CHUNK - 1) // _CHUNK for p in params)
        plan.cap = cap
        plan.eager, plan.spare = _Table(cap, dev), _Table(cap, dev)
        host_steps = [0.0] * len(params)
        for k, p in enumerate(params):
            st = self.state[p]
            o, n = plan.offsets[p], p.numel()
            mv, vv = plan.m[o:o + n].view_as(p), plan.v[o:o + n].view_as(p)
            if "exp_avg" in st:                              # loaded (load_state_dict) or carried-over (add_param_group) state
                mv.copy_(st["exp_avg"])
                vv.copy_(st["exp_avg_sq"])
                host_steps[k] = float(st.get("step", 0))
            st["exp_avg"], st["exp_avg_sq"], st["step"] = mv, vv, plan.steps[k]
        if any(host_steps):
            plan.steps.copy_(torch.tensor(host_steps, dtype=torch.float32))
        self._plans[gi] = plan
        return plan

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._plans = {}                                     # re-adopt the loaded moments into flat buffers at the next step

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        if hasattr(self, "_plans"):
            self._plans.pop(len(self.param_groups) - 1, None)

    # -- capture bookkeeping ---------------------------------------------------------------------------
    def reserve_capture_tables(self):
        """Call OUTSIDE a capture, right before one: every plan gets the table (device bytes + page-locked staging buffer) the
        capture will own.  Page-locked allocations (hipHostMalloc) are not capturable, and a capture consumes the spare without
        a non-capturing step in between to refill it (a re-capture after an lr change in the single-process form, where the
        optimizer step lives inside the graph)."""
        for gi, group in enumerate(self.param_groups):
            plan = self._plan(gi, group)
            if plan is not None and plan.spare is None:
                plan.spare = _Table(plan.cap, plan.m.device)

    def release_captured_tables(self):
        """Call after the graphs that captured this optimizer's step have been destroyed (a re-capture): their tables are
        unreferenced bytes from then on."""
        for plan in self._plans.values():
            plan.captured = []

    # -- device table --------------------------------------------------------------------------------
    def _table(self, plan, group):
        """The table this step's launch reads: the eager one (rebuilt / re-pointed when gradient addresses moved), or a
        fresh one owned by the capture in progress."""
        live = [(p, p.grad) for p in group["params"] if p.requires_grad and p.grad is not None]
        key = tuple((p.data_ptr(), g.data_ptr(), g.numel()) for p, g in live)
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            # owned by this capture and never rewritten: eager steps between replays (a short last batch, validation-time
            # fallbacks) keep using `plan.eager`
            # (no spare: the caller did not `reserve_capture_tables()`; the page-locked allocation below only succeeds in a
            # relaxed-mode capture)
            t, plan.spare = (plan.spare or _Table(plan.cap, plan.m.device)), None
            plan.captured.append(t)
        else:
            t = plan.eager
            if plan.spare is None:
                plan.spare = _Table(plan.cap, plan.m.device)
            if key == t.key:
                return t
        for p, g in live:
            if g.is_sparse:
                raise RuntimeError("Adam does not support sparse gradients, please consider SparseAdam instead")
            if g.dtype != torch.float32 or not g.is_contiguous() or g.device != p.device or g.numel() != p.numel():
                raise RuntimeError("HIP Adam needs contiguous fp32 gradients on the parameter's device")
        ids = tuple(id(p) for p, _ in live)
        same_set = ids == t.live_ids                         # only addresses moved (eager steps re-allocate the gradients)
        if not same_set:
            t.slots = (L.AdamSlot * max(len(live), 1))()
            blocks = []
            for i, (p, g) in enumerate(live):
                n = p.numel()
                t.slots[i].state_off, t.slots[i].n, t.slots[i].step_idx = plan.offsets[p], n, plan.index[p]
                for s0 in range(0, n, _CHUNK):
                    blocks.append((i, min(_CHUNK, n - s0), s0))
            barr = (L.AdamBlock * max(len(blocks), 1))()
            for j, (i, cnt, s0) in enumerate(blocks):
                barr[j].slot, barr[j].count, barr[j].start = i, cnt, s0
            t.nblocks = len(blocks)
        slots = t.slots
        for i, (p, g) in enumerate(live):
            slots[i].p, slots[i].g = p.data_ptr(), g.data_ptr()
        ssz = C.sizeof(L.AdamSlot) * len(live)
        boff = (ssz + 15) // 16 * 16
        total = boff + C.sizeof(L.AdamBlock) * t.nblocks
        upto = total if not same_set else ssz                # the block list depends on the sizes only
        if upto:
            # one host-to-device copy from a page-locked staging buffer sized when the table was made (so nothing is
            # allocated here): captured into a hipGraph it is ONE memcpy node that replays the same bytes
            assert total <= t.host.numel()
            if t.copied is not None and not capturing:
                t.copied.synchronize()                      # the previous table may still be on its way out of `host`
            C.memmove(t.host.data_ptr(), C.addressof(slots), ssz)
            if not same_set:
                C.memmove(t.host.data_ptr() + boff, C.addressof(barr), total - boff)
            t.table[:upto].copy_(t.host[:upto], non_blocking=True)
            if not capturing:
                t.copied = torch.cuda.Event()
                t.copied.record()
        t.key, t.grads, t.live, t.live_ids = key, [g for _, g in live], [p for p, _ in live], ids
        t.nslots, t.blocks_off = len(live), boff
        return t

    # -- step -------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None, grad_scaler=None):
        """grad_scaler: a `DeviceGradScaler` whose scale the gradients still carry (fp16 training): the launch checks them for
        non-finite values, skips the whole update if it finds one, divides by the scale otherwise, and adjusts the scale."""
        on_gpu = {p.is_cuda for group in self.param_groups for p in group["params"]}
        if on_gpu == {False}:
            if grad_scaler is not None:
                raise RuntimeError("DeviceGradScaler needs GPU parameters")
            return super().step(closure)
        if on_gpu != {True}:
            raise RuntimeError("Adam: parameters on both CPU and GPU")
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        launches = []
        for gi, group in enumerate(self.param_groups):
            plan = self._plan(gi, group)
            if plan is None:
                continue
            with torch.cuda.device(plan.m.device):
                t = self._table(plan, group)
                if t.nblocks == 0:
                    continue
                b1, b2 = group["betas"]
                a = L.AdamArgs(slots=t.table.data_ptr(), blocks=t.table.data_ptr() + t.blocks_off,
                               nslots=t.nslots, nblocks=t.nblocks,
                               m=plan.m.data_ptr(), v=plan.v.data_ptr(), steps=plan.steps.data_ptr(), ticket=plan.ticket.data_ptr(),
                               lr=float(group["lr"]), beta1=float(b1), beta2=float(b2), eps=float(group["eps"]),
                               weight_decay=float(group["weight_decay"]), maximize=int(bool(group["maximize"])),
                               one_minus_beta1=1.0 - float(b1), one_minus_beta2=1.0 - float(b2))
                launches.append((plan, t, a))
        # torch.amp.GradScaler.step looks at EVERY group's gradients before it steps any: with several groups all checks come first
        # (an inf in group 1 used to leave group 0 updated); one group keeps the combined entry point (same three launches)
        split = grad_scaler is not None and len(launches) > 1
        if split:
            for plan, t, a in launches:
                with torch.cuda.device(plan.m.device):
                    L.check(L.load().srk_adam_check_scaled(C.byref(a), grad_scaler.state.data_ptr(), torch.cuda.current_stream().cuda_stream), "srk_adam_check_scaled")
        for plan, t, a in launches:
            with torch.cuda.device(plan.m.device):
                stream = torch.cuda.current_stream().cuda_stream
                if grad_scaler is None:
                    L.call("srk_adam_step", a, stream)
                elif split:
                    L.check(L.load().srk_adam_update_scaled(C.byref(a), grad_scaler.state.data_ptr(), stream), "srk_adam_update_scaled")
                else:
                    L.check(L.load().srk_adam_step_scaled(C.byref(a), grad_scaler.state.data_ptr(), stream), "srk_adam_step_scaled")
                # the kernel wrote through raw pointers: tell autograd (and the packed-weight cache, which keys on it) that
                # the parameters changed, as an in-place torch op would
                torch.autograd.graph.increment_version(t.live)
        if grad_scaler is not None:
            with torch.cuda.device(grad_scaler.state.device):
                L.check(L.load().srk_loss_scale_update(grad_scaler.state.data_ptr(), torch.cuda.current_stream().cuda_stream), "srk_loss_scale_update")
        return loss
