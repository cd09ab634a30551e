"""GPU tests of the collapsed HR stage (csrc/hr_tail.hip, ops.HrTailFn): the last `conv3x3 -> PixelShuffle(2)` stage of the upsampler
and the tail conv of EDSR / RCAN / RDN (reference models/common.py:112-139, edsr.py:48-52, rcan.py tail, rdn.py:85-95) as one 5x5
convolution.  Oracle: the reference's two-layer form in float64 torch (tests/collapse_ref.py), on the 16-bit-rounded input."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import collapse_ref as R  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    import sr_amd
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return sr_amd


def _mk(O, C, Ci, seed, bias=True):
    g = torch.Generator().manual_seed(seed)
    wt = (torch.rand(O, C, 3, 3, generator=g) - 0.5) * 0.2
    wu = (torch.rand(4 * C, Ci, 3, 3, generator=g) - 0.5) * 0.1
    bt = (torch.rand(O, generator=g) - 0.5) if bias else None
    bu = (torch.rand(4 * C, generator=g) - 0.5) * 0.2 if bias else None
    return wt, bt, wu, bu


def _relerr(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


@pytest.mark.parametrize("O,C", [(3, 64), (1, 16), (4, 8)])
def test_collapse_kernel_vs_float64(A, O, C):
    """srk_hrtail_collapse: weff / beff and the compact border weights against the float64 construction."""
    L = A._lib
    dev = torch.device("cuda")
    Ci = 64
    wt, bt, wu, bu = _mk(O, C, Ci, 3)
    (We, be), cor = R.collapse(wt.double(), bt.double(), wu.double(), bu.double())
    x = torch.zeros(1, 2, 2, Ci, dtype=torch.bfloat16, device=dev)
    f32 = torch.float32
    bufs = dict(weff=torch.empty((4 * O, Ci, 5, 5), dtype=f32, device=dev), beff=torch.empty(4 * O, dtype=f32, device=dev),
                wedge=torch.empty((4, 2 * O, Ci, 5), dtype=f32, device=dev), bedge=torch.empty((4, 2 * O), dtype=f32, device=dev),
                wcor=torch.empty((4, O, Ci), dtype=f32, device=dev), bcor=torch.empty((4, O), dtype=f32, device=dev))
    d = [t.to(dev).contiguous() for t in (wu, bu, wt, bt)]
    L.call("srk_hrtail_collapse", A.ops.HrTailFn._args(x, d[0], d[1], d[2], d[3], bufs), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    np.testing.assert_allclose(bufs["weff"].cpu().double().numpy(), We.numpy(), rtol=0, atol=2e-6 * float(We.abs().max()))
    np.testing.assert_allclose(bufs["beff"].cpu().double().numpy(), be.numpy(), rtol=0, atol=2e-6 * float(be.abs().max()) + 1e-7)
    # compact border weights: top / bottom keep the row fy = 0 of the (a = 0 / 1) channels, left / right the column fx = 0 of (b = 0 / 1)
    full = {k: (w.view(O, 2, 2, Ci, 5, 5), b.view(O, 2, 2)) for k, (w, b) in cor.items()}
    want_w = torch.stack([full["top"][0][:, 0, :, :, 2, :], full["bot"][0][:, 1, :, :, 2, :],
                          full["left"][0][:, :, 0, :, :, 2], full["right"][0][:, :, 1, :, :, 2]]).reshape(4, 2 * O, Ci, 5)
    want_b = torch.stack([full["top"][1][:, 0, :], full["bot"][1][:, 1, :], full["left"][1][:, :, 0], full["right"][1][:, :, 1]]).reshape(4, 2 * O)
    np.testing.assert_allclose(bufs["wedge"].cpu().double().numpy(), want_w.numpy(), rtol=0, atol=2e-6 * float(want_w.abs().max()))
    np.testing.assert_allclose(bufs["bedge"].cpu().double().numpy(), want_b.numpy(), rtol=0, atol=2e-6 * float(want_b.abs().max()) + 1e-7)
    cw = torch.stack([full[k][0][:, a, b, :, 2, 2] for k, (a, b) in (("tl", (0, 0)), ("tr", (0, 1)), ("bl", (1, 0)), ("br", (1, 1)))])
    cb = torch.stack([full[k][1][:, a, b] for k, (a, b) in (("tl", (0, 0)), ("tr", (0, 1)), ("bl", (1, 0)), ("br", (1, 1)))])
    np.testing.assert_allclose(bufs["wcor"].cpu().double().numpy(), cw.numpy(), rtol=0, atol=2e-6 * float(cw.abs().max()))
    np.testing.assert_allclose(bufs["bcor"].cpu().double().numpy(), cb.numpy(), rtol=0, atol=2e-6 * float(cb.abs().max()) + 1e-7)
    # everything the border terms drop outside these compact slices is multiplied by zero padding: the full tensors vanish there
    assert float(full["top"][0][:, 1].abs().max()) == 0.0 and float(full["left"][0][:, :, 1].abs().max()) == 0.0


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(2, 48, 48), (3, 17, 20), (2, 1, 1), (1, 2, 3), (2, 1, 7), (1, 16, 1), (1, 33, 16)])
@pytest.mark.parametrize("bias", [True, False])
def test_hr_tail_forward_backward_vs_two_layers_float64(A, dt, shape, bias):
    """Forward image, input gradient and all four parameter gradients of ops.hr_tail against the two-layer form in float64 on the
    SAME rounded input: what is left is accumulation order + the 16-bit rounding of the collapsed weights / the un-shuffled gradient
    (and, in the oracle's favour, NO 16-bit rounding of the 64-channel tensor the layer-wise HIP path stores)."""
    n, h, w = shape
    O, C, Ci = 3, 64, 64
    dev = torch.device("cuda")
    wt, bt, wu, bu = _mk(O, C, Ci, 7 + h + w, bias)
    g = torch.Generator().manual_seed(1)
    x = (torch.rand(n, h, w, Ci, generator=g) - 0.5).to(dt)
    post = torch.tensor([0.4488, 0.4371, 0.4040])
    pr = [t.to(dev).requires_grad_(True) if t is not None else None for t in (wu, bu, wt, bt)]
    xd = x.to(dev).requires_grad_(True)
    assert A.ops.hr_tail_ok(xd, pr[0], pr[2], 2)
    y = A.ops.hr_tail(xd, pr[0], pr[1], pr[2], pr[3], post_add=post.to(dev))
    gy = torch.rand(n, O, 2 * h, 2 * w, generator=g) - 0.5
    y.backward(gy.to(dev))
    torch.cuda.synchronize()
    # float64 oracle
    X64 = x.double().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    z = lambda t, s: t.double().requires_grad_(True) if t is not None else torch.zeros(s, dtype=torch.float64)
    Wt, Bt, Wu, Bu = z(wt, None), z(bt, O), z(wu, None), z(bu, 4 * C)
    ref = R.layerwise(X64, Wt, Bt, Wu, Bu) + post.double().view(1, O, 1, 1)
    ref.backward(gy.double())
    tol = 6e-3 if dt == torch.bfloat16 else 1.2e-3
    assert _relerr(y.cpu(), ref.detach()) < tol
    # per element too: the border ring must be as good as the interior (a missing border term is an O(1) error there)
    err = (y.cpu().double() - ref.detach()).abs()
    scale = float(ref.detach().abs().max())
    assert float(err.max()) < 4 * tol * scale, (float(err.max()), scale)
    ring = torch.ones_like(err, dtype=torch.bool)
    if h > 1 and w > 1:
        ring[:, :, 1:-1, 1:-1] = False
    assert float(err[ring].max()) < 4 * tol * scale
    gx_ref = X64.grad.permute(0, 2, 3, 1)
    assert _relerr(xd.grad.cpu(), gx_ref) < (2e-2 if dt == torch.bfloat16 else 3e-3)
    e2 = (xd.grad.cpu().double() - gx_ref).abs()
    assert float(e2.max()) < (0.08 if dt == torch.bfloat16 else 0.02) * float(gx_ref.abs().max())
    for got, want, name in ((pr[0].grad, Wu.grad, "wu"), (pr[2].grad, Wt.grad, "wt")):
        assert _relerr(got.cpu(), want) < (2e-2 if dt == torch.bfloat16 else 3e-3), name
    if bias:
        for got, want, name in ((pr[1].grad, Bu.grad, "bu"), (pr[3].grad, Bt.grad, "bt")):
            assert _relerr(got.cpu(), want) < (2e-2 if dt == torch.bfloat16 else 3e-3), name


def test_hr_tail_equals_layerwise_hip_path_within_16bit_noise(A):
    """The two HIP paths (collapsed / layer by layer, SRK_NO_HR_COLLAPSE) on an EDSR tail: same image within the layer-wise path's own
    16-bit noise, and the collapsed one is the closer of the two to float64."""
    dev = torch.device("cuda")
    torch.manual_seed(0)
    m = A.EDSR(n_feats=64, n_resblocks=2, res_scale=0.1, scale_factor=4, precision="bf16").to(dev)
    x = torch.rand(2, 3, 24, 24, device=dev)
    with torch.no_grad():
        y1 = m(x).float().cpu()
        A.ops._HR_COLLAPSE = False
        try:
            y2 = m(x).float().cpu()
        finally:
            A.ops._HR_COLLAPSE = True
    assert _relerr(y1, y2) < 1e-2
    assert tuple(y1.shape) == (2, 3, 96, 96)


def test_models_take_the_collapsed_path(A, monkeypatch):
    """EDSR x4 / x2, RCAN and RDN in 16-bit call HrTailFn once per forward; fp32 and PixelShuffle(3) keep the layer-wise form."""
    dev = torch.device("cuda")
    calls = []
    orig = A.ops.hr_tail
    monkeypatch.setattr(A.ops, "hr_tail", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    x = torch.rand(1, 3, 12, 12, device=dev)
    with torch.no_grad():
        for mk, want in ((lambda: A.EDSR(n_resblocks=1, scale_factor=4, precision="bf16"), 1),
                         (lambda: A.EDSR(n_resblocks=1, scale_factor=2, precision=16), 1),
                         (lambda: A.RCAN(n_resblocks=1, n_resgroups=1, scale_factor=4, precision="bf16"), 1),
                         (lambda: A.EDSR(n_resblocks=1, scale_factor=3, precision="bf16"), 0),
                         (lambda: A.EDSR(n_resblocks=1, scale_factor=4, precision=32), 0)):
            calls.clear()
            y = mk().to(dev)(x)
            assert len(calls) == want
            assert torch.isfinite(y).all()
