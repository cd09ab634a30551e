"""Minimal fit loop + data-parallel wiring for the `SRModel` surface (Lightning-optional).

The reference has no distributed code of its own: Lightning's Trainer picks DDP when `devices > 1`
(configs/all.yml:83,125-127) and torch DDP all-reduces the fp32 gradients over NCCL (SURVEY.md section 5,
8(e)).  Here: one process per GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI on ROCm) or
"gloo" on CPU, `DistributedDataParallel` with gradients as bucket views and ONE bucket sized to the
model, so that a step issues a single all-reduce that overlaps the tail of backward (xGMI is
point-to-point: fewer, larger messages).  The gradient mean is the only collective on the path.
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
        net = wrap_ddp(model, self.device)
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
            img_sr = net(batch["lr"])                              # DDP hooks the gradient all-reduce here
            result = model._calculate_losses(img_sr=img_sr, img_hr=batch["hr"])
            loss = result["loss"]
            if scaler is not None:
                scaler.scale(loss).backward()
                scaler.step(optimizer)
                scaler.update()
            else:
                loss.backward()
                optimizer.step()
            self.losses.append(float(loss.detach()))
            if self.log_every and self.rank == 0 and (step + 1) % self.log_every == 0:
                print(f"step {step + 1}: loss {self.losses[-1]:.6f}", flush=True)
        return model
