"""Large-kernel convolutions and the collapsed HR stage's forward kernel, element by element on rounded inputs."""


import os


import sys


import numpy as np


import pytest


import torch


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


sys.path.insert(0, ROOT)


from oracle import train as OT  # noqa: E402


pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    import sr_amd
    assert torch.cuda.is_available()
    sr_amd._lib.load()
    return sr_amd


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter((torch.rand(*s, generator=g) - 0.5).cuda()) for s in [(3,), (64, 64, 3, 3), (4097,), (7, 5, 3, 3)]]


def _grads(ps, step, scale):
    g = torch.Generator().manual_seed(77 + step)
    for i, p in enumerate(ps):
        p.grad = ((torch.rand(*p.shape, generator=g) - 0.5) * (10.0 ** (i % 3 - 1)) * scale).to(p.device)


# ---------------------------------------------------------------------------------------------------------------
# the direct large-kernel kernels per ELEMENT against float64 computed from the 16-bit-ROUNDED
# inputs (what is left is accumulation order, <= 1e-3 relative): a packing-permutation slip that touched one (kw, co) pair in 32
# passes a relative-L2 bound of 8 %, not this one
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("k,cout,n,h,w,path", [(9, 3, 2, 40, 33, "lk_wgrad_allrows / lk_conv_rows"), (9, 1, 1, 35, 20, "allrows"), (5, 6, 1, 33, 33, "lk_wgrad_packed"),
                                                (7, 4, 2, 16, 47, "allrows, 7x7"), (7, 16, 2, 17, 30, "lk_wgrad"), (5, 8, 1, 21, 19, "lk5_wgrad"),
                                                (5, 12, 3, 48, 48, "lk5_wgrad (the collapsed HR stage's shape)")])
def test_large_kernel_convs_per_element_on_rounded_inputs(A, dt, k, cout, n, h, w, path):
    import torch.nn.functional as F
    from sr_amd import ops
    g = torch.Generator().manual_seed(17 + k + cout)
    x = (torch.rand(n, 64, h, w, generator=g) * 2 - 1).to(dt)
    wt = (((torch.rand(cout, 64, k, k, generator=g) * 2 - 1) / np.sqrt(64 * k * k)).to(dt)).float()      # weights already representable
    b = (torch.rand(cout, generator=g) * 2 - 1) * 0.1
    gy = (torch.rand(n, cout, h, w, generator=g) * 2 - 1).to(dt)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
    wd, bd = wt.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = ops.conv_general(xd, wd, bd, stride=1, pad=k // 2)
    cp = y.shape[3]
    gyd = torch.zeros(n, h, w, cp, dtype=dt)
    gyd[..., :cout] = gy.permute(0, 2, 3, 1)
    y.backward(gyd.cuda())
    torch.cuda.synchronize()
    x64, w64, b64 = x.double().requires_grad_(True), wt.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = F.conv2d(x64, w64, b64, padding=k // 2)
    ref.backward(gy.double())
    eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    yy = y.detach().float().cpu()[..., :cout].permute(0, 3, 1, 2).double()
    r = ref.detach()
    # forward: one rounding of the stored output per element
    assert float((yy - r).abs().max()) <= 1.1 * eps * float(r.abs().max()) + 1e-6, path
    assert float(y.detach().float()[..., cout:].abs().max()) == 0.0 if cp > cout else True
    gx = xd.grad.float().cpu().permute(0, 3, 1, 2).double()
    assert float((gx - x64.grad).abs().max()) <= 1.1 * eps * float(x64.grad.abs().max()) + 1e-6, path
    # weight / bias gradients: fp32 sums of exact products of 16-bit values
    assert float((wd.grad.cpu().double() - w64.grad).abs().max()) <= 1e-3 * float(w64.grad.abs().max()), path
    assert float((bd.grad.cpu().double() - b64.grad).abs().max()) <= 1e-3 * float(b64.grad.abs().max()) + 1e-5, path


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("O,n,h,w", [(3, 2, 96, 96), (3, 3, 37, 61), (3, 1, 5, 29), (2, 2, 16, 28), (1, 1, 9, 57), (3, 17, 48, 48), (4, 2, 20, 33)])
def test_collapsed_stage_forward_kernel_per_element(A, dt, O, n, h, w):
    """lk5_rows_fwd_kernel ((kernel column, channel) pairs on the MFMA rows, weights in registers, column taps summed through a per-wave
    scratch; O = 4: the tap-per-MFMA kernel) against float64 conv2d + pixel_shuffle on the SAME 16-bit inputs: the image is stored in
    fp32, so what is left is the order of fp32 sums -- a slip in the row / column bookkeeping (band edges, row segments, the ring) is O(1)."""
    import torch.nn.functional as F
    from sr_amd import ops, _lib as L
    g = torch.Generator().manual_seed(5 + O + h + w)
    x = (torch.rand(n, 64, h, w, generator=g) * 2 - 1).to(dt)
    wt = (((torch.rand(4 * O, 64, 5, 5, generator=g) * 2 - 1) / np.sqrt(64 * 25)).to(dt)).float()
    b = (torch.rand(4 * O, generator=g) * 2 - 1) * 0.1
    post = torch.rand(O, generator=g)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    pk = ops.pack_conv(wt.cuda(), b.cuda(), dt, cache=False)
    out = torch.full((n, O, 2 * h, 2 * w), float("nan"), device="cuda")
    ops.conv_raw(xd, pk, N=n, H=h, W=w, Cin=64, Cout=4 * O, out=out, out_mode=L.OUT_PLANAR, ps_r=2, post_add=post.cuda())
    torch.cuda.synchronize()
    ref = F.pixel_shuffle(F.conv2d(x.double(), wt.double(), b.double(), padding=2), 2) + post.double().view(1, O, 1, 1)
    got = out.cpu().double()
    assert bool(torch.isfinite(got).all()), "pixels the kernel never wrote"
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
