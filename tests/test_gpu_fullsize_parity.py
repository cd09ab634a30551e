"""Full-size parity of the HIP path: backward of every model against the reference's own gradient summaries (tests/golden), PSNR of a trained EDSR-baseline within 0.01 dB of the reference path, SRResNet's backward against a float64 oracle."""


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
# full-size backward vs the reference's gradient summaries
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", LARGE)
def test_fullsize_backward_vs_reference_grad_sums(A, name):
    """dL/dparams for L = sum(y * t) of the full-size nets (fp32 path) against [sum, abs-sum, square-sum] of every
    parameter gradient as the REFERENCE computed them (generate_golden.py: grad_summary).  The composed 400-conv
    backward chain of RCAN 10x20 is in here."""
    ent = MANIFEST[name]
    m = getattr(A, ent["class"])(precision=32, **ent["kwargs"])
    fill.formula_fill_module(m)
    m = m.cuda()
    g = np.load(os.path.join(GOLDEN, f"model_{name}.npz"))
    y = m(torch.from_numpy(g["x"]).cuda())
    t = fill.formula_tensor(tuple(y.shape), 77, 1.0).cuda()
    (y * t).sum().backward()
    torch.cuda.synchronize()
    yr = g["y"]
    assert float(np.abs(y.detach().cpu().numpy() - yr).max()) <= 1e-3 * max(1.0, float(np.abs(yr).max()))
    params = dict(m.named_parameters())
    names = [str(n) for n in g["grad_names"]]
    assert len(names) == sum(1 for p in m.parameters() if p.requires_grad)
    # scale on which tiny gradients are judged: the largest abs-sum per element in the net
    worst = 0.0
    for n, s in zip(names, g["grad_sums"]):
        gr = params[n].grad.double().flatten().cpu()
        got = np.array([float(gr.sum()), float(gr.abs().sum()), float((gr * gr).sum())])
        # abs-sum and square-sum are well conditioned; the plain sum can cancel (cf. tests/test_oracle_golden.py)
        e1 = abs(got[1] - s[1]) / max(s[1], 1e-30)
        e2 = abs(got[2] - s[2]) / max(s[2], 1e-30)
        e0 = abs(got[0] - s[0]) / max(s[1], 1e-30)
        worst = max(worst, e0, e1, e2)
        assert e1 < 4e-2 and e2 < 8e-2 and e0 < 4e-2, f"{name} grad {n}: sum {e0:.2e} abs-sum {e1:.2e} square-sum {e2:.2e}"
    print(f"{name}: worst relative deviation of a gradient summary {worst:.2e} over {len(names)} tensors")


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


@pytest.mark.parametrize("prec", ["bf16", 16, 32])
def test_psnr_within_0p01_db_of_reference_path(A, trained_edsr, prec):
    kw, sd, lr_t, hr_t, y_ref = trained_edsr
    m = A.EDSR(precision=prec, **kw)
    m.load_state_dict(sd)
    m = m.cuda().eval()
    with torch.no_grad():
        y = m(lr_t.cuda()).float().cpu()
    p_ref, p_build = psnr(y_ref, hr_t), psnr(y, hr_t)
    p_cross = psnr(y, y_ref)
    print(f"precision {prec}: PSNR(oracle, hr) {p_ref:.3f} dB, PSNR(build, hr) {p_build:.3f} dB, delta {p_build - p_ref:+.4f} dB, "
          f"PSNR(build, oracle) {p_cross:.1f} dB, max|err| {float((y - y_ref).abs().max()):.2e}")
    assert p_ref > 20.0, "the trained net should actually super-resolve the smooth images"
    per = [psnr(y[i:i + 1], hr_t[i:i + 1]) - psnr(y_ref[i:i + 1], hr_t[i:i + 1]) for i in range(hr_t.shape[0])]
    print("   per-image deltas (dB): " + " ".join(f"{d:+.4f}" for d in per))
    # fp32 / fp16 storage: within 0.01 dB, data-set mean and image by image.  A RAW bf16 forward (8 mantissa bits on the
    # 16-block residual trunk) measures 0.005-0.013 dB below the reference path: bounded here at 0.03 dB; the evaluation
    # entry points of a bf16 model (validation_step / predict_step) therefore run in fp16 storage -- next assertion
    lim = 0.03 if prec == "bf16" else 0.01
    assert abs(p_build - p_ref) < lim and max(abs(d) for d in per) < lim, (p_build - p_ref, per)
    with torch.no_grad():
        ye = m.predict_step({"lr": lr_t.cuda()}, 0).float().cpu()
    pe = [psnr(ye[i:i + 1], hr_t[i:i + 1]) - psnr(y_ref[i:i + 1], hr_t[i:i + 1]) for i in range(hr_t.shape[0])]
    print(f"   predict_step (eval dtype {m.eval_dtype}): mean delta {psnr(ye, hr_t) - p_ref:+.4f} dB, per image " + " ".join(f"{d:+.4f}" for d in pe))
    assert abs(psnr(ye, hr_t) - p_ref) < 0.01 and max(abs(d) for d in pe) < 0.01, pe


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


def test_srresnet_fullsize_backward_vs_float64_oracle(A):
    """Full-size SRResNet (16 blocks, 64 features, x4) forward + backward, fp32 HIP path vs the oracle evaluated in float64.
    Bound: what fp32 arithmetic itself achieves on this ill-conditioned net (torch CPU fp32 vs float64: up to 4 % relative
    L2 on a tensor) with margin; every sizeable gradient must also point the same way (cosine)."""
    from oracle import init as OI
    ent = MANIFEST["srresnet_full_x4"]
    m = A.SRResNet(precision=32, **ent["kwargs"])
    fill.formula_fill_module(m)
    m = m.cuda()
    g = np.load(os.path.join(GOLDEN, "model_srresnet_full_x4.npz"))
    x = torch.from_numpy(g["x"])
    y = m(x.cuda())
    t = fill.formula_tensor(tuple(y.shape), 77, 1.0)
    (y * t.cuda()).sum().backward()
    sd, tr = OI.build_state_dict("SRResNet", **ent["kwargs"])
    fill.formula_fill_state_dict(sd, tr)
    for k in list(sd):
        if sd[k].is_floating_point():
            sd[k].data = sd[k].data.double()
    for k in tr:
        sd[k].requires_grad_(True)
    yr = OF.forward("SRResNet", sd, x.double(), **ent["kwargs"])
    (yr * t.double()).sum().backward()
    assert float((y.detach().cpu().double() - yr.detach()).abs().max()) <= 1e-3 * max(1.0, float(yr.abs().max()))
    gmax = max(float(sd[k].grad.norm()) for k in tr)
    worst = 0.0
    for n, p in m.named_parameters():
        ref = sd[n].grad.flatten()
        if float(ref.norm()) < 1e-4 * gmax:
            continue                                   # conv biases in front of a BatchNorm: their true gradient is zero
        got = p.grad.double().flatten().cpu()
        e = float((got - ref).norm() / ref.norm())
        cos = float(torch.dot(got, ref) / (got.norm() * ref.norm()))
        worst = max(worst, e)
        assert e < 0.1 and cos > 0.99, f"{n}: rel-L2 {e:.3f} cosine {cos:.4f}"
    print(f"srresnet_full_x4: worst relative L2 deviation of a gradient from the float64 oracle {worst:.3f}")
