"""Metric core and the L1 loss on the device (csrc/data.hip).  Part of `ops` (re-exported there): split out of ops.py in round 6."""
import os

import torch

from . import _lib as L
from .ops import _f32c, _need_gpu, _stream      # (ops.py imports this module at its END: these exist by then)


# --------------------------------------------------------------------------------------------
# metric core on the device
# --------------------------------------------------------------------------------------------
def image_sse(sr, hr, *, luma=False, shave=0):
    """Per-image sum of squared differences of clamp(sr,0,1) and clamp(hr,0,1) (all channels, or BT.601 luma) with a
    `shave`-pixel border removed, and the element count per image.  srmodel.py:224-232,582 (piq.psnr core)."""
    _need_gpu(sr)
    n, c, h, w = sr.shape
    s32, h32 = _f32c(sr), _f32c(hr)
    sse = torch.zeros(n, dtype=torch.float64, device=sr.device)
    a = L.SseArgs(sr=s32.data_ptr(), hr=h32.data_ptr(), N=n, C=c, H=h, W=w, luma=int(luma), shave=int(shave), sse=sse.data_ptr())
    L.call("srk_image_sse", a, _stream())
    count = (h - 2 * shave) * (w - 2 * shave) * (1 if luma else c)
    return sse, count


def psnr(sr, hr, *, luma=False, shave=0, eps=1e-8):
    """10 log10(1 / (MSE + eps)) per image, batch mean (piq.psnr defaults: data_range 1, EPS 1e-8)."""
    sse, count = image_sse(sr, hr, luma=luma, shave=shave)
    return (10.0 * torch.log10(1.0 / (sse / count + eps))).mean().float()


def ssim(x, y, *, sigma=1.5, k1=0.01, k2=0.03):
    """SSIM with piq.ssim's defaults on the device (srk_image_ssim): average-pool by max(1, round(min(H, W) / 256)),
    separable 11-tap Gaussian, mean of the valid SSIM map per (image, channel), then mean over channels and images.
    No host synchronisation: the result is a 0-d device tensor."""
    _need_gpu(x)
    xs, ys = _f32c(x), _f32c(y)
    n, c, h, w = xs.shape
    f = max(1, round(min(h, w) / 256))
    sums = torch.zeros(n * c, dtype=torch.float64, device=xs.device)
    a = L.SsimArgs(x=xs.data_ptr(), y=ys.data_ptr(), N=n, C=c, H=h, W=w, pool=f, sigma=float(sigma), k1=float(k1), k2=float(k2),
                   sums=sums.data_ptr())
    L.call("srk_image_ssim", a, _stream())
    count = (h // f - 10) * (w // f - 10)
    return (sums / count).mean().float()


class L1LossFn(torch.autograd.Function):
    """mean |sr - hr| (F.l1_loss, reference srmodel.py:160-171) as two HIP launches per step instead of torch's
    sub / abs / mean / sign / mul chain: forward reads both images once and keeps sign(sr - hr) as int8, backward
    expands the signs into the gradient (no second read of the images, no host sync: gout stays on the device)."""

    @staticmethod
    def forward(ctx, sr, hr):
        _need_gpu(sr)
        s, h = _f32c(sr), _f32c(hr)
        n = s.numel()
        sign = torch.empty(n, dtype=torch.int8, device=s.device)
        nb = L.load().srk_l1_blocks(n)
        partial = torch.empty(nb, dtype=torch.float64, device=s.device)
        a = L.L1Args(sr=s.data_ptr(), hr=h.data_ptr(), n=n, sign=sign.data_ptr(), partial=partial.data_ptr(), gout=0, scale=0.0, grad=0)
        L.call("srk_l1_loss_fwd", a, _stream())
        ctx.save_for_backward(sign)
        ctx.shape = tuple(sr.shape)
        out = torch.empty((), dtype=torch.float32, device=s.device)
        L.check(L.load().srk_l1_loss_mean(partial.data_ptr(), nb, n, out.data_ptr(), _stream()), "srk_l1_loss_mean")
        return out

    @staticmethod
    def backward(ctx, g):
        (sign,) = ctx.saved_tensors
        n = sign.numel()
        gout = g.detach().float().contiguous()
        grad = torch.empty(ctx.shape, dtype=torch.float32, device=sign.device)
        a = L.L1Args(sr=0, hr=0, n=n, sign=sign.data_ptr(), partial=0, gout=gout.data_ptr(), scale=1.0 / n, grad=grad.data_ptr())
        L.call("srk_l1_loss_bwd", a, _stream())
        return grad, None


def l1_loss(sr, hr):
    """F.l1_loss(sr, hr) for device tensors (hr needs no gradient)."""
    if hr.requires_grad or sr.numel() == 0:
        return torch.nn.functional.l1_loss(sr, hr)
    return L1LossFn.apply(sr, hr)
