"""Minimal fit loop + data-parallel wiring for the `SRModel` surface (Lightning-optional).

The reference has no distributed code of its own: Lightning's Trainer picks DDP when `devices > 1`
(configs/all.yml:83,125-127) and torch DDP all-reduces the fp32 gradients over NCCL (SURVEY.md section 5,
8(e)).  Here: one process per GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI on ROCm) or
"gloo" on CPU, and ONE all-reduce of a flat gradient buffer per step (`GradSync`; xGMI is point-to-point: fewer,
larger messages).  `wrap_ddp` (DistributedDataParallel, one bucket sized to the model, bucket views) is kept as the
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


def wrap_ddp(model, device, force=False):
    """DDP wrapper tuned for this path: one bucket (all grads), bucket views, no unused-parameter scan."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force):
        return model
    nbytes = sum(p.numel() * 4 for p in model.parameters() if p.requires_grad)
    cap_mb = max(1, int(nbytes / 2 ** 20) + 1)
    kw = dict(gradient_as_bucket_view=True, bucket_cap_mb=cap_mb, broadcast_buffers=False, find_unused_parameters=False)
    if device.type == "cuda":
        return DDP(model, device_ids=[device.index], output_device=device.index, **kw)
    return DDP(model, **kw)


class GradSync:
    """The path's only collective, without the DDP wrapper: after backward the fp32 gradients are gathered into ONE flat
    buffer (one multi-tensor copy), averaged with ONE all-reduce (RCCL over xGMI on the GPUs, gloo on CPU) and handed
    back to the parameters as views of that buffer.  Same arithmetic as DistributedDataParallel with a single bucket
    (mean over ranks); what it drops is the per-parameter autograd hooks and the reducer's bookkeeping, which cost
    0.65 ms of a 9 ms EDSR-baseline step on one rank.  Nothing overlaps with backward, by design: the largest model
    here (EDSR-large, 172 MB of gradients) is a ~1 ms all-reduce against a 44 ms step.
    Replicas start identical: `broadcast()` sends rank 0's parameters and buffers."""

    def __init__(self, module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        self.module = module
        self.flat = None
        self.views = None
        self.world = dist.get_world_size() if dist.is_initialized() else 1

    def broadcast(self):
        if self.world > 1:
            with torch.no_grad():
                for t in list(self.module.parameters()) + list(self.module.buffers()):
                    dist.broadcast(t, src=0)

    def sync(self):
        """Average the gradients over the ranks (call between backward and the optimizer step)."""
        if not dist.is_initialized():
            return
        live = [(p, p.grad) for p in self.params if p.grad is not None]
        if not live:
            return
        total = sum(g.numel() for _, g in live)
        if self.flat is None or self.flat.numel() != total or self.flat.device != live[0][1].device:
            self.flat = torch.empty(total, dtype=torch.float32, device=live[0][1].device)
        views, off = [], 0
        for _, g in live:
            views.append(self.flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        torch._foreach_copy_(views, [g for _, g in live])
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        if self.world > 1:
            self.flat.mul_(1.0 / self.world)
        for (p, _), v in zip(live, views):
            p.grad = v


def synthetic_batch(n, channels, lr_size, scale, seed, device):
    """The BASELINE workload: uniform [0,1) LR patches and HR targets."""
    g = torch.Generator().manual_seed(seed)
    lr = torch.rand(n, channels, lr_size, lr_size, generator=g)
    hr = torch.rand(n, channels, lr_size * scale, lr_size * scale, generator=g)
    return {"lr": lr.to(device), "hr": hr.to(device), "path": [f"synthetic/{i}" for i in range(n)]}


class Trainer:
    """`fit(model, batches)`: forward -> losses -> backward -> optimizer.step, DDP when WORLD_SIZE > 1.

    Mirrors what Lightning's fit loop does around `SRModel.training_step` (srmodel.py:160-171):
    nothing else (no checkpointing / loggers -- out of scope)."""

    def __init__(self, device=None, max_steps=-1, log_every=0, use_grad_scaler=None):
        if device is None:
            device = torch.device("cuda", dist_env()[2]) if torch.cuda.is_available() else torch.device("cpu")
        self.device = torch.device(device)
        self.max_steps = max_steps
        self.log_every = log_every
        self.use_grad_scaler = use_grad_scaler
        self.rank, self.world, self.local = init_distributed(self.device.type)
        self.losses = []

    def fit(self, model, batches):
        model = model.to(self.device)
        use_ddp = os.environ.get("SRK_USE_TORCH_DDP") == "1"
        net = wrap_ddp(model, self.device) if use_ddp else model
        gsync = None
        if not use_ddp and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            gsync = GradSync(model)
            gsync.broadcast()
        optimizer = model.configure_optimizers()[0]
        scaler = None
        use_scaler = self.use_grad_scaler
        if use_scaler is None:
            use_scaler = getattr(model, "compute_dtype", torch.float32) == torch.float16 and self.device.type == "cuda"
        if use_scaler:
            scaler = torch.amp.GradScaler("cuda")
        for step, batch in enumerate(batches):
            if 0 <= self.max_steps <= step:
                break
            batch = {k: (v.to(self.device) if torch.is_tensor(v) else v) for k, v in batch.items()}
            optimizer.zero_grad(set_to_none=True)
            img_sr = net(batch["lr"])
            result = model._calculate_losses(img_sr=img_sr, img_hr=batch["hr"])
            loss = result["loss"]
            if scaler is not None:
                scaler.scale(loss).backward()
                if gsync is not None:
                    gsync.sync()
                scaler.step(optimizer)
                scaler.update()
            else:
                loss.backward()
                if gsync is not None:
                    gsync.sync()
                optimizer.step()
            self.losses.append(float(loss.detach()))
            if self.log_every and self.rank == 0 and (step + 1) % self.log_every == 0:
                print(f"step {step + 1}: loss {self.losses[-1]:.6f}", flush=True)
        return model
