"""Unfold / fold (a conv and its adjoint) and the BatchNorm / PReLU kernels against torch."""


import json


import math


import os


import socket


import subprocess


import sys


import numpy as np


import pytest


import torch


import torch.nn.functional as F


from oracle import fill, functional as OF


pytestmark = pytest.mark.gpu


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


MANIFEST = json.load(open(os.path.join(GOLDEN, "manifest.json")))


# srresnet_full_x4 is judged against the float64 oracle instead (test_srresnet_fullsize_backward_vs_float64_oracle): with
# 33 BatchNorm layers over 288 pixels its gradients are ill-conditioned -- the reference's own fp32 result is 4 % (relative
# L2, worst tensor) away from the float64 value of the same expression
LARGE = sorted(k for k, v in MANIFEST.items() if v["class"] != "SRCNN" and v["n_params_trainable"] >= 1_000_000 and k != "srresnet_full_x4")


@pytest.fixture(scope="module")
def A():
    import sr_amd
    assert torch.cuda.is_available()
    return sr_amd


# ---------------------------------------------------------------------------------------------------------------
# PSNR within 0.01 dB of the reference path
# ---------------------------------------------------------------------------------------------------------------
def smooth_images(n, size, seed):
    """Smooth synthetic 'photographs': sums of low-frequency sin*cos products per channel plus a little noise, in [0,1]."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, size), torch.linspace(0, 1, size), indexing="ij")
    out = torch.zeros(n, 3, size, size)
    for i in range(n):
        for c in range(3):
            img = torch.zeros(size, size)
            for _ in range(6):
                fx, fy = (torch.rand(2, generator=g) * 9 + 0.5).tolist()
                px, py = (torch.rand(2, generator=g) * 6.28).tolist()
                amp = float(torch.rand(1, generator=g)) * 0.25
                img += amp * torch.sin(6.28 * fx * xx + px) * torch.cos(6.28 * fy * yy + py)
            out[i, c] = 0.5 + img
    out += 0.01 * torch.randn(out.shape, generator=g)
    return out.clamp(0, 1)


def psnr(a, b):
    mse = ((a.double().clamp(0, 1) - b.double().clamp(0, 1)) ** 2).flatten(1).mean(1)
    return float((10.0 * torch.log10(1.0 / (mse + 1e-12))).mean())


@pytest.fixture(scope="module")
def trained_edsr(A):
    """EDSR-baseline x4 trained for 300 Adam steps (bf16 HIP path) on smooth 192x192 images, bicubic LR."""
    kw = dict(n_feats=64, n_resblocks=16, res_scale=0.1, scale_factor=4)
    torch.manual_seed(0)
    m = A.EDSR(precision="bf16", **kw).cuda()
    hr = smooth_images(48, 192, 11)
    lr = F.interpolate(hr, scale_factor=0.25, mode="bicubic", antialias=True).clamp(0, 1)
    hr_d, lr_d = hr.cuda(), lr.cuda()
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3)      # the reference's effective optimizer
    g = torch.Generator().manual_seed(1)
    first = last = None
    for step in range(300):
        idx = torch.randint(0, 48, (16,), generator=g).cuda()
        opt.zero_grad(set_to_none=True)
        loss = m.training_step({"lr": lr_d[idx], "hr": hr_d[idx]}, step)["loss"]
        loss.backward()
        opt.step()
        first = float(loss) if first is None else first
        last = float(loss)
    assert math.isfinite(last) and last < 0.5 * first, (first, last)
    sd = {k: v.detach().float().cpu().clone() for k, v in m.state_dict().items()}
    hr_t = smooth_images(6, 192, 99)
    lr_t = F.interpolate(hr_t, scale_factor=0.25, mode="bicubic", antialias=True).clamp(0, 1)
    with torch.no_grad():
        y_ref = OF.forward("EDSR", sd, lr_t, **kw)
    return kw, sd, lr_t, hr_t, y_ref


# ---------------------------------------------------------------------------------------------------------------
# two models alternating in one process (BASELINE configs[4])
# ---------------------------------------------------------------------------------------------------------------
def _steps(A, cls, kw, batches, other=None):
    torch.manual_seed(0)
    m = getattr(A, cls)(precision=16, **kw).cuda()
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-4, fused=True)
    losses = []
    for b in batches:
        opt.zero_grad(set_to_none=True)
        loss = m.training_step(b, 0)["loss"]
        (loss * 128.0).backward()
        for p in m.parameters():
            if p.grad is not None:
                p.grad.mul_(1.0 / 128.0)
        opt.step()
        losses.append(float(loss))
        if other is not None:
            other()
    return losses, {k: v.detach().float().cpu().clone() for k, v in m.state_dict().items()}


# ---------------------------------------------------------------------------------------------------------------
# RCCL path: GradSync on a 1-rank nccl group (2 ranks when there are 2 GPUs)
# ---------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


DDP_WORKER = r"""
import os, sys, torch
sys.path.insert(0, {root!r})
import sr_amd
from sr_amd import trainer as T
rank, world, local = T.init_distributed("cuda", force=True)
dev = torch.device("cuda", local)
torch.manual_seed(0)
mode = {mode!r}
m = sr_amd.EDSR(n_feats=64, n_resblocks=4 if mode == "segments" else 2, res_scale=0.1, scale_factor=2, precision="bf16").to(dev)
g = torch.Generator().manual_seed(5)
full = [{{"lr": torch.rand(4, 3, 24, 24, generator=g), "hr": torch.rand(4, 3, 48, 48, generator=g)}} for _ in range(6)]
per = 4 // world
gs = T.GradSync(m, overlap=(mode != "pack_reduce"), bucket_bytes=64 << 10)
gs.broadcast()
opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, fused=True)
if mode == "graphed":
    # the trainer's loop: eager steps with the overlapped all-reduces, then forward + backward + packing as one hipGraph, the
    # all-reduce issued eagerly, the optimizer step (the one-launch HIP Adam) as a second graph
    opt = m.configure_optimizers()[0]
    gstep = T.GraphedStep(m, m, opt, gs, warm_steps=2)
    for b in full:
        gstep({{k: v[rank * per:(rank + 1) * per].to(dev) for k, v in b.items()}})
    assert gstep.graphs is not None and len(gstep.graphs) == 2 and not gstep.failed
    gstep.finish()          # (the last replay's update: the multi-rank graph opens with the optimizer step of the previous one)
    full = []
if mode == "segments":
    # large-model form: the backward pass as three graph segments with the bucket all-reduces issued between them
    os.environ["SRK_DDP_SEGMENTS"] = "3"
    opt = m.configure_optimizers()[0]
    gstep = T.GraphedStep(m, m, opt, gs, warm_steps=2)
    for b in full:
        gstep({{k: v[rank * per:(rank + 1) * per].to(dev) for k, v in b.items()}})
    assert gstep.ogs is not None and gstep.ogs.nseg == 3 and len(gstep.graphs) == 4 and not gstep.failed, (gstep.ogs and gstep.ogs.nseg, gstep.failed)
    assert len(gstep.ogs.gsync.group_buckets) >= 3 and all(gstep.ogs.gsync.group_buckets[k] for k in range(3))
    full = []
for b in full:
    sh = {{k: v[rank * per:(rank + 1) * per].to(dev) for k, v in b.items()}}
    opt.zero_grad(set_to_none=True)
    m._calculate_losses(img_sr=m(sh["lr"]), img_hr=sh["hr"])["loss"].backward()
    if mode == "hooks":
        gs.sync()
    else:
        gs.pack(); gs.reduce()
    assert all(p.grad.data_ptr() == gs.views[p].data_ptr() for p in gs.params)
    opt.step()
torch.cuda.synchronize()
torch.save({{k: v.float().cpu() for k, v in m.state_dict().items()}}, os.path.join({out!r}, f"{{mode}}_r{{rank}}.pt"))
torch.distributed.barrier()
torch.distributed.destroy_process_group()
"""


@pytest.mark.parametrize("shape", [(2, 10, 9, 16, 3, 1, 1), (1, 12, 12, 32, 8, 4, 2), (2, 8, 6, 16, 6, 2, 2), (1, 5, 7, 32, 9, 1, 4), (1, 9, 9, 16, 12, 8, 2)])
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_unfold_fold_are_conv_and_its_adjoint(A, shape, dt):
    """srk_unfold_nhwc + 1x1 conv == F.conv2d(stride, padding); 1x1 conv + srk_fold_nhwc == F.conv_transpose2d; both with
    input / weight / bias gradients (float64 reference from inputs rounded to the compute dtype)."""
    n, h, w, c, k, st, pd = shape
    if (h + 2 * pd - k) < 0:
        pytest.skip("kernel larger than the padded image")
    cout = 32
    g = torch.Generator().manual_seed(k * 100 + st)
    x = (torch.rand(n, c, h, w, generator=g) - 0.5)
    wc = (torch.rand(cout, c, k, k, generator=g) - 0.5) / (k * c ** 0.5)
    wt = (torch.rand(c, cout, k, k, generator=g) - 0.5) / (k * c ** 0.5)
    b = torch.rand(cout, generator=g) - 0.5
    tol = 2e-4 if dt == torch.float32 else 4e-2
    q = lambda t: t.to(dt).double()
    for mode in ("conv", "deconv"):
        xd = x.permute(0, 2, 3, 1).contiguous().to(dt).cuda().requires_grad_(True)
        wp = torch.nn.Parameter((wc if mode == "conv" else wt).clone().cuda())
        bp = torch.nn.Parameter(b.clone().cuda())
        xr = q(x).requires_grad_(True)
        wr = q(wc if mode == "conv" else wt).requires_grad_(True)
        br = b.double().requires_grad_(True)
        if mode == "conv":
            y = A.ops.conv_general(xd, wp, bp, stride=st, pad=pd)
            yr = F.conv2d(xr, wr, br, stride=st, padding=pd)
        else:
            y = A.ops.conv_transpose_general(xd, wp, bp, stride=st, pad=pd)
            yr = F.conv_transpose2d(xr, wr, br, stride=st, padding=pd)
        assert tuple(y.shape) == (n, yr.shape[2], yr.shape[3], cout)
        t = torch.rand(yr.shape, generator=g, dtype=torch.float64) - 0.5
        (y.float() * t.permute(0, 2, 3, 1).float().cuda()).sum().backward()
        (yr * t).sum().backward()
        torch.cuda.synchronize()
        rel = lambda got, ref: float((got.double().cpu() - ref).abs().max() / max(1e-9, float(ref.abs().max())))
        assert rel(y.detach().permute(0, 3, 1, 2), yr.detach()) < tol, mode
        assert rel(xd.grad.permute(0, 3, 1, 2), xr.grad) < tol, mode
        assert rel(wp.grad, wr.grad) < tol, mode
        assert rel(bp.grad, br.grad) < tol, mode


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_batchnorm_and_prelu_kernels(A, dt):
    """srk_chan_stats / srk_chan_apply as nn.BatchNorm2d (train + eval, with a fused residual) and nn.PReLU (shared and
    per-channel slope): values, input gradients and parameter gradients vs float64 torch."""
    g = torch.Generator().manual_seed(3)
    n, c, h, w = 3, 32, 9, 7
    tol = 1e-4 if dt == torch.float32 else 3e-2
    x = torch.randn(n, c, h, w, generator=g) * 1.5 + 0.3
    r = torch.randn(n, c, h, w, generator=g)
    t = torch.randn(n, c, h, w, generator=g).double()
    q = lambda v: v.to(dt).double()
    rel = lambda got, ref: float((got.double().cpu() - ref).abs().max() / max(1e-9, float(ref.abs().max())))
    for training in (True, False):
        bn = torch.nn.BatchNorm2d(c).cuda()
        ref = torch.nn.BatchNorm2d(c).double()
        with torch.no_grad():
            for m_ in (bn, ref):
                m_.weight.copy_(torch.linspace(0.5, 1.5, c)); m_.bias.copy_(torch.linspace(-0.2, 0.2, c))
                m_.running_mean.copy_(torch.linspace(-0.1, 0.4, c)); m_.running_var.copy_(torch.linspace(0.8, 2.0, c))
        bn.train(training); ref.train(training)
        xd = x.permute(0, 2, 3, 1).contiguous().to(dt).cuda().requires_grad_(True)
        rd = r.permute(0, 2, 3, 1).contiguous().to(dt).cuda().requires_grad_(True)
        y = A.ops.batch_norm(xd, bn, res=rd)
        xr, rr = q(x).requires_grad_(True), q(r).requires_grad_(True)
        yr = ref(xr) + rr
        (y.float() * t.permute(0, 2, 3, 1).float().cuda()).sum().backward()
        (yr * t).sum().backward()
        assert rel(y.detach().permute(0, 3, 1, 2), yr.detach()) < tol
        assert rel(xd.grad.permute(0, 3, 1, 2), xr.grad) < 4 * tol and rel(rd.grad.permute(0, 3, 1, 2), rr.grad) < tol
        assert rel(bn.weight.grad, ref.weight.grad) < 4 * tol and rel(bn.bias.grad, ref.bias.grad) < 4 * tol
        assert rel(bn.running_mean, ref.running_mean) < tol and rel(bn.running_var, ref.running_var) < tol
        assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked)
    for npar in (1, c):
        a = torch.nn.Parameter(torch.linspace(0.05, 0.4, npar).cuda())
        ar = torch.linspace(0.05, 0.4, npar).double().requires_grad_(True)
        xd = x.permute(0, 2, 3, 1).contiguous().to(dt).cuda().requires_grad_(True)
        y = A.ops.prelu(xd, a)
        xr = q(x).requires_grad_(True)
        yr = F.prelu(xr, ar)
        (y.float() * t.permute(0, 2, 3, 1).float().cuda()).sum().backward()
        (yr * t).sum().backward()
        assert rel(y.detach().permute(0, 3, 1, 2), yr.detach()) < tol and rel(xd.grad.permute(0, 3, 1, 2), xr.grad) < tol
        assert rel(a.grad, ar.grad) < 4 * tol
