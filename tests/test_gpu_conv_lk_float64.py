"""The direct large-kernel convolutions (csrc/conv_lk.hip) against float64."""


import numpy as np


import pytest


import torch


from oracle import functional as OF, train as OT


pytestmark = pytest.mark.gpu


PREC = {torch.float16: 16, torch.bfloat16: "bf16"}


@pytest.fixture(scope="module")
def A():
    import sr_amd
    assert torch.cuda.is_available()
    sr_amd._lib.load()
    return sr_amd


RCAN_KW = dict(n_feats=64, n_resgroups=2, n_resblocks=3, reduction=16, scale_factor=2)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("k,cout,n,h,w", [(9, 3, 2, 40, 33), (9, 3, 1, 16, 16), (5, 8, 1, 21, 19), (7, 16, 2, 17, 30),
                                         # few real output channels: the weight gradient with (kw, co) pairs on the MFMA columns (cout * k <= 32)
                                         (9, 1, 1, 35, 20), (7, 4, 2, 16, 47), (5, 6, 1, 33, 33)])
def test_direct_large_kernel_conv_vs_float64(A, dt, k, cout, n, h, w):
    """ops.conv_general on the direct 5x5 / 7x7 / 9x9 kernels (csrc/conv_lk.hip: SRResNet's 9x9 tail conv 64 -> 3, models/srresnet.py:29)
    against float64 F.conv2d: output, data gradient, weight and bias gradients (no column tensor: the forward must not call
    srk_unfold_nhwc)."""
    import torch.nn.functional as F
    from sr_amd import ops
    tol, l2 = ({torch.bfloat16: 4e-2, torch.float16: 6e-3}[dt], {torch.bfloat16: 8e-2, torch.float16: 3e-2}[dt])
    g = torch.Generator().manual_seed(5)
    x = torch.rand(n, 64, h, w, generator=g) * 2 - 1
    wt = (torch.rand(cout, 64, k, k, generator=g) * 2 - 1) / np.sqrt(64 * k * k)
    b = (torch.rand(cout, generator=g) * 2 - 1) * 0.1
    gy = torch.rand(n, cout, h, w, generator=g) * 2 - 1
    xr, wr, br = x.to(dt).double().requires_grad_(True), wt.to(dt).double().requires_grad_(True), b.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, padding=k // 2)
    yr.backward(gy.to(dt).double())
    xd = x.permute(0, 2, 3, 1).contiguous().to(dt).cuda().requires_grad_(True)
    wd, bd = torch.nn.Parameter(wt.cuda()), torch.nn.Parameter(b.cuda())
    calls = []
    opj = A.ops_proj                      # (unfold / fold and the general strided convs live in ops_proj since round 6)
    orig = opj._unfold_raw
    opj._unfold_raw = lambda *a_, **k_: (calls.append(1), orig(*a_, **k_))[1]
    try:
        y = ops.conv_general(xd, wd, bd, stride=1, pad=k // 2)
        gyd = torch.zeros(n, h, w, y.shape[3], dtype=dt)
        gyd[..., :cout] = gy.permute(0, 2, 3, 1).to(dt)
        y.backward(gyd.cuda())
    finally:
        opj._unfold_raw = orig
    torch.cuda.synchronize()
    assert not calls, "the large-kernel conv went through im2col"

    def rel(a_, b_):
        return float((a_.double().cpu() - b_).abs().max() / b_.abs().max())

    def l2e(a_, b_):
        return float((a_.double().cpu() - b_).norm() / b_.norm())
    assert rel(y.detach()[..., :cout].permute(0, 3, 1, 2), yr.detach()) < tol
    assert float(y.detach()[..., cout:].abs().max()) == 0.0 if y.shape[3] > cout else True
    assert l2e(xd.grad.permute(0, 3, 1, 2), xr.grad) < l2
    assert l2e(wd.grad, wr.grad) < l2
    assert l2e(bd.grad, br.grad) < l2
