"""The models' reference-facing surface on the GPU: several models in one process, stand-alone blocks and their NCHW contract, NaN propagation to the loss, eval mode / running statistics, predict.py."""


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


def test_two_models_alternating_match_each_alone(A):
    """WDSR-B and RDN (fp16) take training steps alternately in one process: each must follow exactly the trajectory it
    follows alone (packed-weight groups, deferred weight-gradient queue and thread-local state are per model / per
    pass; nothing leaks between the two)."""
    g = torch.Generator().manual_seed(8)
    bw = [{"lr": torch.rand(2, 3, 24, 24, generator=g).cuda(), "hr": torch.rand(2, 3, 96, 96, generator=g).cuda()} for _ in range(3)]
    br = [{"lr": torch.rand(2, 3, 20, 28, generator=g).cuda(), "hr": torch.rand(2, 3, 80, 112, generator=g).cuda()} for _ in range(3)]
    kw_w, kw_r = dict(type="B", scale_factor=4, n_resblocks=4), dict(rdn_config="A", scale_factor=4)
    lw, sw = _steps(A, "WDSR", kw_w, bw)
    lr_, sr = _steps(A, "RDN", kw_r, br)
    # interleaved: an RDN step runs between every two WDSR steps (and vice versa)
    torch.manual_seed(0)
    rdn = A.RDN(precision=16, **kw_r).cuda()
    ropt = torch.optim.Adam([p for p in rdn.parameters() if p.requires_grad], lr=1e-4, fused=True)
    it = iter(br)
    rl = []

    def rdn_step():
        b = next(it)
        ropt.zero_grad(set_to_none=True)
        loss = rdn.training_step(b, 0)["loss"]
        (loss * 128.0).backward()
        for p in rdn.parameters():
            if p.grad is not None:
                p.grad.mul_(1.0 / 128.0)
        ropt.step()
        rl.append(float(loss))
    lw2, sw2 = _steps(A, "WDSR", kw_w, bw, other=rdn_step)
    sr2 = {k: v.detach().float().cpu() for k, v in rdn.state_dict().items()}
    assert lw2 == lw and rl == lr_, (lw, lw2, lr_, rl)
    for k in sw:
        assert torch.equal(sw[k], sw2[k]), k
    for k in sr:
        assert torch.equal(sr[k], sr2[k]), k


def test_standalone_block_after_model_forward_packs_fresh_weights(A):
    """a block called on its own after a model's forward must not be served the model's packed weights
    (stale after an optimizer step): the packed group only serves inside its forward window."""
    torch.manual_seed(0)
    m = A.EDSR(n_feats=64, n_resblocks=2, scale_factor=2, precision="bf16").cuda()
    x = torch.rand(1, 3, 16, 16, device="cuda")
    m(x)
    blk = m.body[0]
    f = (torch.rand(1, 16, 16, 64, device="cuda") - 0.5).to(torch.bfloat16)
    with torch.no_grad():
        y0 = blk.nhwc(f).float()
        for p in blk.parameters():
            p.mul_(0.0)                      # an "optimizer step" that does not bump what the group keys on
        y1 = blk.nhwc(f).float()
    assert float((y1 - f.float()).abs().max()) == 0.0, "zeroed weights: the block must return its input"
    assert float((y0 - f.float()).abs().max()) > 0.0


def test_blocks_keep_the_reference_nchw_contract(A):
    """`common.py` blocks composed on NCHW tensors by a user's SRModel subclass (reference
    README.md:97-101) -- forward(x NCHW) -> NCHW, values and gradients as torch's own ops give them."""
    import torch.nn.functional as F_
    from sr_amd.models import common as C
    torch.manual_seed(0)
    conv = C.DefaultConv2d(in_channels=8, out_channels=24, kernel_size=3).cuda()
    blk = C.ResBlock(n_feats=32, res_scale=0.5).cuda()
    up = C.UpscaleBlock(2, 16).cuda()
    ms = C.MeanShift().cuda()
    x = torch.rand(2, 8, 11, 13, device="cuda", requires_grad=True)
    y = conv(x)
    ref = F_.conv2d(x.detach(), conv.weight, conv.bias, padding=1)
    assert tuple(y.shape) == (2, 24, 11, 13) and y.dtype == torch.float32
    assert float((y - ref).abs().max()) < 1e-4
    y.square().sum().backward()
    xr = x.detach().clone().requires_grad_(True)
    F_.conv2d(xr, conv.weight.detach(), conv.bias.detach(), padding=1).square().sum().backward()
    assert float((x.grad - xr.grad).abs().max()) < 1e-3 * float(xr.grad.abs().max())
    f = torch.rand(1, 32, 9, 9, device="cuda")
    c1, c2 = blk.body[0], blk.body[2]
    rb = F_.conv2d(F_.relu(F_.conv2d(f, c1.weight, c1.bias, padding=1)), c2.weight, c2.bias, padding=1) * 0.5 + f
    assert float((blk(f) - rb).abs().max()) < 1e-4
    u = torch.rand(1, 16, 6, 7, device="cuda")
    ru = F_.pixel_shuffle(F_.conv2d(u, up[0].weight, up[0].bias, padding=1), 2)
    assert tuple(up(u).shape) == (1, 16, 12, 14) and float((up(u) - ru).abs().max()) < 1e-4
    img = torch.rand(1, 3, 5, 5, device="cuda")
    assert float((ms(img) - F_.conv2d(img, ms.weight, ms.bias)).abs().max()) < 1e-6


def test_nan_input_reaches_the_loss(A):
    """Documented deviation: the fused ReLU is max(x, 0) with IEEE maxNum semantics, so a NaN
    pre-activation becomes 0 where torch.relu would keep it.  A non-finite activation still reaches the output (and the
    loss, and GradScaler's inf check) through the skip connections."""
    torch.manual_seed(0)
    m = A.EDSR(n_feats=64, n_resblocks=2, scale_factor=2, precision=16).cuda()
    x = torch.rand(1, 3, 16, 16, device="cuda")
    x[0, 1, 5, 7] = float("nan")
    with torch.no_grad():
        y = m(x)
    assert not bool(torch.isfinite(y).all())


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


# ---------------------------------------------------------------------------------------------------------------
# predict.py on a saved checkpoint
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.timeout(900)
def test_predict_script_writes_pngs(A, tmp_path):
    from PIL import Image
    torch.manual_seed(0)
    kw = dict(n_feats=64, n_resblocks=2, res_scale=0.1, scale_factor=4)
    m = A.EDSR(precision="bf16", **kw)
    ck = tmp_path / "edsr.ckpt"
    torch.save({"state_dict": m.state_dict()}, ck)
    d = tmp_path / "Set5"
    d.mkdir()
    rng = np.random.default_rng(0)
    img = rng.integers(0, 255, (37, 45, 3), dtype=np.uint8)
    Image.fromarray(img).save(d / "bird.png")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "predict.py"), "-m", "edsr", "--checkpoint", str(ck), "--n_feats", "64", "--n_resblocks", "2", "--res_scale", "0.1",
                          "--predict_datasets", str(d), "--default_root_dir", str(tmp_path / "res")],
                         capture_output=True, text=True, timeout=800)
    if out.returncode != 0 and "unrecognized arguments" in out.stderr:
        pytest.fail(out.stderr[-1500:])
    assert out.returncode == 0, out.stderr[-3000:]
    got = np.asarray(Image.open(tmp_path / "res" / "Set5" / "bird.png"))
    assert got.shape == (148, 180, 3)
    m = m.cuda().eval()
    with torch.no_grad():
        sr = m.predict_step({"lr": torch.from_numpy(img.copy()).permute(2, 0, 1).float()[None].cuda() / 255.0}, 0)
    want = m.to_uint8(sr[0]).permute(1, 2, 0).cpu().numpy()
    assert np.array_equal(got, want)
    assert (tmp_path / "res" / "Set5" / "bird_center.png").exists()


# ---------------------------------------------------------------------------------------------------------------
# SRResNet / DDBPN
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", sorted(k for k, v in MANIFEST.items() if v["class"] == "SRResNet"))
def test_srresnet_eval_mode_and_running_statistics(A, name):
    """BatchNorm: one training-mode forward updates the running buffers like the reference's modules did, and the
    eval-mode forward that follows reproduces the reference's eval output (fixture: y_eval, buffer_sums)."""
    ent = MANIFEST[name]
    m = A.SRResNet(precision=32, **ent["kwargs"])
    fill.formula_fill_module(m)
    m = m.cuda()
    g = np.load(os.path.join(GOLDEN, f"model_{name}.npz"))
    x = torch.from_numpy(g["x"]).cuda()
    with torch.no_grad():
        y = m(x)
        m.eval()
        ye = m(x)
    torch.cuda.synchronize()
    assert float(np.abs(y.cpu().numpy() - g["y"]).max()) <= 1e-3 * max(1.0, float(np.abs(g["y"]).max()))
    assert float(np.abs(ye.cpu().numpy() - g["y_eval"]).max()) <= 1e-3 * max(1.0, float(np.abs(g["y_eval"]).max()))
    bufs = dict(m.named_buffers())
    for n, s3 in zip([str(v) for v in g["buffer_names"]], g["buffer_sums"]):
        b = bufs[n].double().flatten().cpu()
        np.testing.assert_allclose([float(b.sum()), float(b.abs().sum()), float((b * b).sum())], s3, rtol=2e-3, atol=1e-6, err_msg=n)
