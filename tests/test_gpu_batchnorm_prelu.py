"""BatchNorm + PReLU: gradient accumulation, one-pass statistics, the fused unit against float64, a shared instance used twice."""


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


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_batchnorm_prelu_gradients_accumulate_into_existing_grads(A, dt):
    """ops.batch_norm / ops.prelu (nn.BatchNorm2d + nn.PReLU of SRResNet's blocks, models/srresnet.py:16-21): the statistics launch
    also finalizes (srk_chan_stats_finalize), counts num_batches_tracked and, when the parameters already HAVE fp32 gradients, adds
    dgamma / dbeta / dslope into them in place.  Two backward passes (None -> tensors from autograd, then in-place accumulation)
    against torch in float64 on the same NCHW data."""
    from sr_amd import ops
    torch.manual_seed(3)
    n, c, h, w = 4, 64, 12, 10
    bn = torch.nn.BatchNorm2d(c).cuda()
    pr = torch.nn.PReLU(c, init=0.2).cuda()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
    bn_r = torch.nn.BatchNorm2d(c).double()
    pr_r = torch.nn.PReLU(c, init=0.2).double()
    bn_r.load_state_dict({k: v.detach().cpu().double() if v.is_floating_point() else v.detach().cpu() for k, v in bn.state_dict().items()})
    tol = 3e-2 if dt == torch.bfloat16 else 2e-4
    for it in range(2):
        x = torch.randn(n, c, h, w) * 2 + 1
        gy = torch.randn(n, c, h, w)
        xq = x.to(dt)
        xd = xq.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
        y = ops.prelu(ops.batch_norm(xd, bn), pr.weight)
        y.backward(gy.to(dt).permute(0, 2, 3, 1).contiguous().cuda())
        xr = xq.double().requires_grad_(True)
        yr = pr_r(bn_r(xr))
        yr.backward(gy.to(dt).double())
        torch.cuda.synchronize()
        assert int(bn.num_batches_tracked) == it + 1 == int(bn_r.num_batches_tracked)
        for got, ref, name in [(bn.weight.grad, bn_r.weight.grad, "dgamma"), (bn.bias.grad, bn_r.bias.grad, "dbeta"), (pr.weight.grad, pr_r.weight.grad, "dslope"),
                               (xd.grad.permute(0, 3, 1, 2), xr.grad, "dx")]:
            err = float((got.double().cpu() - ref).norm() / ref.norm())
            assert err < tol, f"pass {it}: {name} off by {err:.2e}"
        assert float((bn.running_var.double().cpu() - bn_r.running_var).abs().max()) < tol


@pytest.mark.parametrize("first_pixel", ["typical", "outlier"])
def test_batchnorm_one_pass_statistics_are_well_conditioned(A, first_pixel):
    """|mean| >> std in fp32 (300 +- 0.05: E[x^2] - mean^2 would lose every digit of the variance): the one-pass statistics of
    ops.batch_norm shift the data by the tensor's first pixel before summing (srk_chan_stats shift_out), so the variance keeps
    ~4 digits -- also when that pixel is 20 standard deviations out.  Against float64."""
    from sr_amd import ops
    torch.manual_seed(5)
    n, c, h, w = 8, 64, 24, 24
    x = 300.0 + 0.05 * torch.randn(n, h, w, c)
    if first_pixel == "outlier":
        x[0, 0, 0, :] = 301.0
    bn = torch.nn.BatchNorm2d(c).cuda()
    y = ops.batch_norm(x.cuda().requires_grad_(True), bn)
    xr = x.double().view(-1, c)
    mean, var = xr.mean(0), xr.var(0, unbiased=False)
    yr = ((xr - mean) / torch.sqrt(var + bn.eps)).view(n, h, w, c)
    torch.cuda.synchronize()
    assert float((bn.running_mean.double().cpu() - 0.1 * mean).abs().max()) < 1e-4
    assert float((bn.running_var.double().cpu() - (0.9 + 0.1 * xr.var(0, unbiased=True))).abs().max() / 0.9) < 1e-5
    err = float((y.detach().double().cpu() - yr).abs().max())
    assert err < 2e-2, f"normalised output off by {err} (values are ~N(0, 1); the fp32 input has 24 bits for 300 +- 0.05: ~6e-4 of a sigma)"


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("npar", [1, 32])
@pytest.mark.parametrize("shape", [(3, 32, 9, 7), (4, 32, 48, 48)])          # few blocks (fused finalize) / 18 blocks
def test_fused_batchnorm_prelu_vs_float64(A, dt, npar, shape):
    """ops.batch_norm_prelu (SRResNet's conv -> BatchNorm -> PReLU, srresnet.py:16-21 via common.py:94-100, as one unit: the BatchNorm
    output is recomputed in the backward pass, its statistics and the slope's gradient come from ONE pass) against float64
    F.batch_norm + F.prelu: values, input gradient, the gradients of gamma / beta / slope, running buffers, num_batches_tracked;
    a second backward accumulates into the existing gradients."""
    import torch.nn.functional as F
    from sr_amd import ops
    g = torch.Generator().manual_seed(13)
    n, c, h, w = shape
    tol = {torch.float32: 2e-4, torch.bfloat16: 3e-2, torch.float16: 4e-3}[dt]
    x = torch.randn(n, c, h, w, generator=g) * 1.5 + 0.3
    t = torch.randn(n, c, h, w, generator=g)
    q = lambda v: v.to(dt).double()
    rel = lambda got, ref: float((got.double().cpu() - ref).abs().max() / max(1e-9, float(ref.abs().max())))
    bn = torch.nn.BatchNorm2d(c).cuda()
    ref = torch.nn.BatchNorm2d(c).double()
    with torch.no_grad():
        for m_ in (bn, ref):
            m_.weight.copy_(torch.linspace(0.5, 1.5, c)); m_.bias.copy_(torch.linspace(-0.4, 0.2, c))
    a = torch.nn.Parameter(torch.linspace(0.05, 0.4, npar).cuda())
    ar = torch.linspace(0.05, 0.4, npar).double().requires_grad_(True)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dt).cuda().requires_grad_(True)
    td = t.permute(0, 2, 3, 1).contiguous().to(dt).cuda()
    y = ops.batch_norm_prelu(xd, bn, a)
    assert type(y.grad_fn).__name__ == "BNPReLUFnBackward"
    xr = q(x).requires_grad_(True)
    yr = F.prelu(ref(xr), ar)
    y.backward(td)
    yr.backward(q(t))
    assert rel(y.detach().permute(0, 3, 1, 2), yr.detach()) < tol
    assert rel(xd.grad.permute(0, 3, 1, 2), xr.grad) < 6 * tol
    assert rel(bn.weight.grad, ref.weight.grad) < 4 * tol and rel(bn.bias.grad, ref.bias.grad) < 4 * tol
    assert rel(a.grad, ar.grad) < 4 * tol
    assert rel(bn.running_mean, ref.running_mean) < tol and rel(bn.running_var, ref.running_var) < tol
    assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked) == 1
    first = [p.grad.clone() for p in (bn.weight, bn.bias, a)]
    ptrs = [p.grad.data_ptr() for p in (bn.weight, bn.bias, a)]
    y2 = ops.batch_norm_prelu(xd, bn, a)
    y2.backward(td)
    for p, f, ptr in zip((bn.weight, bn.bias, a), first, ptrs):
        assert p.grad.data_ptr() == ptr
        torch.testing.assert_close(p.grad, 2 * f, rtol=1e-5, atol=1e-5)


def test_shared_batchnorm_prelu_instance_used_twice_in_one_forward(A):
    """The reference's ResBlock appends the SAME BatchNorm2d / PReLU instance behind both of its convs (common.py:94-100; SRResNet):
    the second use of a backward pass adds its [C]-sized gradients into the tensor the first use handed to autograd (ops._pass_slot).
    Against float64 torch, two passes in a row (the second with existing .grad buffers: accumulation)."""
    import torch.nn.functional as F
    from sr_amd import ops
    g = torch.Generator().manual_seed(21)
    c = 32
    x = torch.randn(2, c, 10, 9, generator=g)
    t = torch.randn(2, c, 10, 9, generator=g)
    bn, ref = torch.nn.BatchNorm2d(c).cuda(), torch.nn.BatchNorm2d(c).double()
    with torch.no_grad():
        for m_ in (bn, ref):
            m_.weight.copy_(torch.linspace(0.5, 1.5, c)); m_.bias.copy_(torch.linspace(-0.3, 0.3, c))
    a = torch.nn.Parameter(torch.full((c,), 0.25).cuda())
    ar = torch.full((c,), 0.25).double().requires_grad_(True)
    rel = lambda got, want: float((got.double().cpu() - want).abs().max() / max(1e-9, float(want.abs().max())))
    for rounds in (1, 2):
        xd = x.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)            # fp32 storage
        h = ops.batch_norm_prelu(xd, bn, a)                                            # use 1: BatchNorm + PReLU
        y = ops.batch_norm(h * 0.5 + 0.1, bn)                                          # use 2 of the same BatchNorm (no activation)
        y = ops.prelu(y, a)                                                            # use 2 of the same PReLU
        y.backward(t.permute(0, 2, 3, 1).contiguous().cuda())
        xr = x.double().requires_grad_(True)
        hr = F.prelu(ref(xr), ar)
        yr = F.prelu(ref(hr * 0.5 + 0.1), ar)
        yr.backward(t.double())
        assert rel(xd.grad.permute(0, 3, 1, 2), xr.grad) < 2e-3
        assert rel(bn.weight.grad, ref.weight.grad) < 2e-3 and rel(bn.bias.grad, ref.bias.grad) < 2e-3 and rel(a.grad, ar.grad) < 2e-3
