"""PSNR / SSIM / uint8 rounding restatements (oracle: test infrastructure).  PARITY UNPINNED.

The reference computes its metrics with `piq.psnr` / `piq.ssim` at default
arguments on clamped RGB float tensors (models/srmodel.py:52-53,224-232,582) and
writes PNGs with `torchvision.utils.save_image` (srmodel.py:311-315,409-412).
piq (pinned 0.7.0, Dockerfile_fixed_versions:63) and torchvision are NOT
installable in the build container and the reference holds no tests or golden
values for them, so these functions restate the published algorithms:

  psnr      10 log10(data_range^2 / MSE) per image over C,H,W, mean over the batch
            (piq.psnr defaults: data_range=1.0, reduction='mean', convert_to_greyscale=False)
  ssim      Wang et al. 2004, 11x11 Gaussian sigma 1.5, k1=0.01, k2=0.03, valid window,
            per-channel then mean; inputs average-pooled by f=max(1, round(min(H,W)/256))
            (piq.ssim defaults)
  psnr_y    BT.601 luma, border shave = scale (the community SR convention; BASELINE.json
            says "PSNR-Y", the reference itself computes RGB PSNR -- both are reported)
  to_uint8  floor(clamp(x,0,1)*255 + 0.5)  (torchvision save_image: mul(255).add_(0.5).clamp_(0,255).to(uint8))
"""
import torch
import torch.nn.functional as F


def psnr(x, y, data_range=1.0):
    mse = ((x.double() - y.double()) ** 2).flatten(1).mean(dim=1)
    return (10.0 * torch.log10(data_range ** 2 / (mse + 1e-8))).mean()   # piq adds EPS=1e-8


def rgb_to_y(x):
    """BT.601 luma in [16/255, 235/255] for x in [0,1], NCHW with 3 channels."""
    r, g, b = x[:, 0:1], x[:, 1:2], x[:, 2:3]
    return (65.481 * r + 128.553 * g + 24.966 * b + 16.0) / 255.0


def psnr_y(x, y, shave):
    xy, yy = rgb_to_y(x.double()), rgb_to_y(y.double())
    if shave > 0:
        xy, yy = xy[..., shave:-shave, shave:-shave], yy[..., shave:-shave, shave:-shave]
    mse = ((xy - yy) ** 2).flatten(1).mean(dim=1)
    return (10.0 * torch.log10(1.0 / mse.clamp_min(1e-12))).mean()


def _gauss_kernel(size=11, sigma=1.5, dtype=torch.float64):
    c = torch.arange(size, dtype=dtype) - (size - 1) / 2.0
    g = torch.exp(-(c ** 2) / (2 * sigma ** 2))
    g = g / g.sum()
    return torch.outer(g, g)


def ssim(x, y, data_range=1.0, k1=0.01, k2=0.03, kernel_size=11, sigma=1.5):
    x, y = x.double() / data_range, y.double() / data_range
    f = max(1, round(min(x.shape[-2:]) / 256))
    if f > 1:
        x, y = F.avg_pool2d(x, f), F.avg_pool2d(y, f)
    c = x.shape[1]
    k = _gauss_kernel(kernel_size, sigma).view(1, 1, kernel_size, kernel_size).repeat(c, 1, 1, 1)
    c1, c2 = k1 ** 2, k2 ** 2
    mu_x, mu_y = F.conv2d(x, k, groups=c), F.conv2d(y, k, groups=c)
    sxx = F.conv2d(x * x, k, groups=c) - mu_x ** 2
    syy = F.conv2d(y * y, k, groups=c) - mu_y ** 2
    sxy = F.conv2d(x * y, k, groups=c) - mu_x * mu_y
    cs = (2 * sxy + c2) / (sxx + syy + c2)
    ss = (2 * mu_x * mu_y + c1) / (mu_x ** 2 + mu_y ** 2 + c1) * cs
    return ss.mean(dim=(-1, -2)).mean(dim=1).mean()


def to_uint8(x):
    return torch.floor(x.clamp(0, 1) * 255.0 + 0.5).to(torch.uint8)
