"""GPU tests added in round 2 (VERDICT r1 "what's weak" 1, 3, 10d and "missing" 4):

* north_star's "PSNR within 0.01 dB of reference": a TRAINED EDSR-baseline on smooth images, HIP bf16 / fp16 forward vs
  the fp32 CPU oracle with the same weights, PSNR of both against the common HR target;
* full-size backward (EDSR-baseline / -large, RCAN 10x20, RDN-A/B, WDSR-A/B) against the reference's own gradient
  summaries in tests/golden (fp32 path);
* two models alternating in one process (BASELINE configs[4]: WDSR-B + RDN, fp16);
* the RCCL path in the GPU suite: a HIP model under trainer.GradSync on a 1-rank `nccl` group (and 2 ranks when the
  box has 2 GPUs);
* predict.py on a saved checkpoint: PNG out with save_image's rounding.
"""
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


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode", ["hooks", "pack_reduce", "graphed", "segments"])
def test_gradsync_over_rccl(A, tmp_path, mode):
    """A HIP EDSR trains 6 steps under trainer.GradSync on an `nccl` (= RCCL) process group: 1 rank always (the
    collective path itself), 2 ranks when the box has 2 GPUs (replica equality).  The result must equal the same
    steps without any process group (global batch)."""
    ngpu = torch.cuda.device_count()
    world = 2 if ngpu >= 2 else 1
    script = tmp_path / "worker.py"
    script.write_text(DDP_WORKER.format(root=ROOT, mode=mode, out=str(tmp_path)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=800, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    sds = [torch.load(tmp_path / f"{mode}_r{r}.pt") for r in range(world)]
    for k in sds[0]:
        for s in sds[1:]:
            assert torch.equal(sds[0][k], s[k]), f"replicas diverged at {k}"
    # the same three steps in this process, no process group
    torch.manual_seed(0)
    m = A.EDSR(n_feats=64, n_resblocks=4 if mode == "segments" else 2, res_scale=0.1, scale_factor=2, precision="bf16").cuda()
    g = torch.Generator().manual_seed(5)
    full = [{"lr": torch.rand(4, 3, 24, 24, generator=g), "hr": torch.rand(4, 3, 48, 48, generator=g)} for _ in range(6)]
    opt = m.configure_optimizers()[0] if mode in ("graphed", "segments") else torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3, fused=True)
    for b in full:
        opt.zero_grad(set_to_none=True)
        m._calculate_losses(img_sr=m(b["lr"].cuda()), img_hr=b["hr"].cuda())["loss"].backward()
        opt.step()
    for k, v in m.state_dict().items():
        dv = (v.float().cpu() - sds[0][k]).abs()
        if mode in ("graphed", "segments"):
            # replayed steps vs launch-by-launch steps: the same kernels, but fp32 atomics of the small weight gradients and
            # Adam's lr-sized moves on near-zero gradients let single weights part by a few steps' worth
            assert float(dv.max()) <= 6.5e-3 and float(dv.mean()) <= 3e-4, (k, float(dv.max()), float(dv.mean()))
        else:
            assert float(dv.max()) <= (2e-6 if world == 1 else 2e-4) * max(1.0, float(v.abs().max())), (k, float(dv.max()))


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


def test_trainer_short_batch_right_after_the_capture_then_replays():
    """the eager fallback for a batch of another shape (4 -> the first batch after the capture) is followed by
    six replays; with fp32 storage the replayed loop must stay on the eager loop's trajectory step for step (a replay that
    applied the short batch's stale gradients would not), and an lr change after the capture must reach the replays."""
    import sr_amd
    from sr_amd import trainer as T

    def batches():
        for i in range(12):
            n = 8 if i != 4 else 3                       # steps 0-2 eager, 3 = capture + first replay, 4 = short batch
            yield T.synthetic_batch(n, 3, 24, 2, 300 + i, "cpu")

    out = []
    for use_graph in (True, False):
        torch.manual_seed(0)
        m = sr_amd.EDSR(scale_factor=2, precision=32, n_feats=32, n_resblocks=2, res_scale=0.1)
        tr = T.Trainer(device="cuda", use_graph=use_graph)

        class Sched:                                     # a manual "scheduler": lr drops after step 8
            def __iter__(self_):
                for i, b in enumerate(batches()):
                    if i == 8:
                        for grp in self_.opt.param_groups:
                            grp["lr"] = 2.5e-4
                    yield b
        sch = Sched()
        orig = m.configure_optimizers

        def conf():
            r = orig()
            sch.opt = r[0]
            return r
        m.configure_optimizers = conf
        tr.fit(m, sch)
        torch.cuda.synchronize()
        out.append((tr.losses, [p.detach().clone() for p in m.parameters()], tr.graphed))
    (lg, pg, g), (le, pe, _) = out
    assert g is not None and g.graphs is not None and not g.failed
    assert len(lg) == len(le) == 12
    np.testing.assert_allclose(lg, le, rtol=2e-4)
    for a, b in zip(pg, pe):
        assert float((a - b).abs().max()) <= 2e-4, float((a - b).abs().max())


@pytest.mark.parametrize("model_kw", [dict(cls="EDSR", n_feats=64, n_resblocks=2, res_scale=0.1), dict(cls="RCAN", n_feats=64, n_resblocks=3, n_resgroups=2)])
def test_trainer_graph_replay_follows_the_eager_loop(model_kw):
    """Trainer.fit replays the training step as a hipGraph after three eager steps (trainer.GraphedStep): same losses, step for
    step, and the same final weights as the launch-by-launch loop; a batch of another shape falls back to eager."""
    import sr_amd
    from sr_amd import trainer as T
    kw = dict(model_kw)
    cls = getattr(sr_amd, kw.pop("cls"))

    def batches():
        for i in range(9):
            n = 16 if i != 7 else 6                      # one short batch in the middle of the replays
            yield T.synthetic_batch(n, 3, 24, 2, 100 + i, "cpu")

    out = []
    for use_graph in (True, False):
        torch.manual_seed(0)
        m = cls(scale_factor=2, precision="bf16", **kw)
        tr = T.Trainer(device="cuda", use_graph=use_graph)
        tr.fit(m, batches())
        torch.cuda.synchronize()
        out.append((tr.losses, [p.detach().clone() for p in m.parameters()], tr.graphed))
    (lg, pg, g), (le, pe, _) = out
    assert g is not None and g.graphs is not None and not g.failed, "the step was captured"
    assert len(lg) == len(le) == 9
    np.testing.assert_allclose(lg, le, rtol=2e-3)
    assert lg[0] == le[0], "the first step is the same eager code on the same weights"
    np.testing.assert_allclose(lg[:3], le[:3], rtol=1e-4)
    # Adam moves a weight by up to lr = 1e-3 per step whatever the size of its gradient, so single weights with near-zero
    # gradients may part by a few steps' worth; on average the two runs stay together
    for a, b in zip(pg, pe):
        assert float((a - b).abs().max()) <= 9.5e-3 and float((a - b).abs().mean()) <= 3e-4
