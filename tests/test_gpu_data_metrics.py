"""GPU parity of the steps either side of the path against the oracle:
srk_sample_patches vs oracle.data (itself pinned to PIL), srk_image_sse vs oracle.metrics."""
import random

import numpy as np
import pytest
import torch

from oracle import data as OD, metrics as OM

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scale,patch,channels", [(4, 48, 3), (2, 32, 3), (3, 24, 1), (4, 192, 3)])
def test_patch_sampler_bit_exact(scale, patch, channels):
    import sr_amd as A
    rng = np.random.default_rng(5)
    pairs = []
    for h, w in ((70, 96), (64, 64), (51, 83)):
        h, w = max(h, patch // scale + 3), max(w, patch // scale + 5)
        pairs.append((rng.integers(0, 256, (h, w, channels), dtype=np.uint8),
                      rng.integers(0, 256, (h * scale, w * scale, channels), dtype=np.uint8)))
    s = A.data.PatchSampler(pairs, scale, patch, augment=True)
    pyrng = random.Random(11)
    idx = [pyrng.randrange(len(pairs)) for _ in range(24)]
    params = [s.draw(i, pyrng) for i in idx]
    assert {p[2] for p in params} == {0, 90, 180, 270} and {p[3] for p in params} == {True, False}
    out = s.batch(idx, params)
    torch.cuda.synchronize()
    assert out["lr"].dtype == torch.float32 and tuple(out["hr"].shape) == (24, channels, patch, patch)
    for k, (i, (top, left, angle, hf, vf)) in enumerate(zip(idx, params)):
        lo, ho = OD.get_patch_pair(pairs[i][0], pairs[i][1], top, left, patch // scale, scale, angle, hf, vf)
        np.testing.assert_array_equal(out["lr"][k].cpu().numpy(), lo)
        np.testing.assert_array_equal(out["hr"][k].cpu().numpy(), ho)


def test_patch_sampler_feeds_training_step():
    import sr_amd as A
    rng = np.random.default_rng(1)
    pairs = [(rng.integers(0, 256, (40, 40, 3), dtype=np.uint8), rng.integers(0, 256, (160, 160, 3), dtype=np.uint8))]
    s = A.data.PatchSampler(pairs, 4, 96)
    m = A.EDSR(n_feats=16, n_resblocks=1, precision="bf16").cuda()
    res = m.training_step(s.batch([0, 0, 0, 0], rng=random.Random(2)), 0)
    res["loss"].backward()
    assert torch.isfinite(res["loss"])


@pytest.mark.parametrize("n,h,w", [(1, 37, 53), (3, 192, 192), (2, 340, 492)])
def test_psnr_reductions(n, h, w):
    import sr_amd as A
    g = torch.Generator().manual_seed(3)
    hr = torch.rand(n, 3, h, w, generator=g) * 1.2 - 0.1           # some values outside [0,1]: the clamp matters
    sr = hr + 0.05 * torch.randn(n, 3, h, w, generator=g)
    got = float(A.ops.psnr(sr.cuda(), hr.cuda()))
    ref = float(OM.psnr(sr.clamp(0, 1), hr.clamp(0, 1)))
    assert abs(got - ref) < 1e-4, (got, ref)
    goty = float(A.ops.psnr(sr.cuda(), hr.cuda(), luma=True, shave=4, eps=0.0))
    refy = float(OM.psnr_y(sr.clamp(0, 1), hr.clamp(0, 1), 4))
    assert abs(goty - refy) < 1e-3, (goty, refy)
    m = A.EDSR(n_feats=16, n_resblocks=1, metrics=["PSNR", "SSIM", "PSNR-Y"])
    out = m._calculate_metrics(sr.clamp(0, 1).cuda(), hr.clamp(0, 1).cuda(), 1)
    assert abs(float(out["Set5/PSNR"]) - ref) < 1e-4 and abs(float(out["Set5/PSNR-Y"]) - refy) < 1e-3


@pytest.mark.parametrize("n,c,h,w", [(1, 3, 37, 53), (2, 3, 192, 192), (1, 1, 11, 11), (2, 3, 600, 530), (1, 3, 1020, 768)])
def test_ssim_device_reduction(n, c, h, w):
    """srk_image_ssim against the float64 oracle (piq.ssim's published defaults): structured + noisy images, the pooled
    path (min(H, W) >= 384 -> factor >= 2, extents not divisible by the factor), a single-window image."""
    import sr_amd as A
    g = torch.Generator().manual_seed(h * 7 + w)
    yy, xx = torch.meshgrid(torch.linspace(0, 6.28, h), torch.linspace(0, 9.42, w), indexing="ij")
    base = (0.5 + 0.25 * torch.sin(yy)[None, None] * torch.cos(xx)[None, None]).expand(n, c, h, w)
    hr = (base + 0.1 * torch.rand(n, c, h, w, generator=g)).clamp(0, 1).contiguous()
    sr = (hr + 0.05 * torch.randn(n, c, h, w, generator=g)).clamp(0, 1).contiguous()
    ref = float(OM.ssim(sr, hr))
    got = float(A.ops.ssim(sr.cuda(), hr.cuda()))
    assert abs(got - ref) < 2e-5, (got, ref)
    assert abs(float(A.ops.ssim(hr.cuda(), hr.cuda())) - 1.0) < 1e-6
    if c == 3:
        m = A.EDSR(n_feats=16, n_resblocks=1, metrics=["SSIM"])
        out = m._calculate_metrics(sr.cuda(), hr.cuda(), 1)
        assert abs(float(out["Set5/SSIM"]) - ref) < 2e-5


@pytest.mark.parametrize("shape", [(2, 3, 48, 48), (1, 3, 37, 53), (3, 1, 5, 7), (4, 3, 192, 192)])
def test_l1_loss_fused(shape):
    """srk_l1_loss_fwd / _bwd against F.l1_loss: value (fixed-order double partials) and gradient, incl. exact zeros
    (sr == hr somewhere: sign 0), sizes that are not a multiple of 4, an upstream gradient other than 1."""
    import sr_amd as A
    g = torch.Generator().manual_seed(sum(shape))
    hr = torch.rand(shape, generator=g)
    sr = (hr + 0.1 * torch.randn(shape, generator=g))
    sr.view(-1)[::7] = hr.view(-1)[::7]
    a = sr.clone().cuda().requires_grad_(True)
    b = sr.clone().double().requires_grad_(True)
    la = A.ops.l1_loss(a, hr.cuda())
    lb = torch.nn.functional.l1_loss(b, hr.double())
    (la * 0.37).backward()
    (lb * 0.37).backward()
    assert abs(float(la) - float(lb)) < 1e-6 * max(1.0, abs(float(lb)))
    assert torch.equal(a.grad.cpu() == 0, b.grad == 0)
    assert float((a.grad.cpu().double() - b.grad).abs().max()) < 1e-7 * float(b.grad.abs().max())
    la2 = A.ops.l1_loss(a.detach(), hr.cuda())
    assert float(la2) == float(la), "fixed-order reduction: bitwise reproducible"
