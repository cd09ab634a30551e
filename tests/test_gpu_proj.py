"""GPU parity of D-DBPN's direct projection kernels (csrc/proj.hip: srk_proj_pack / srk_proj_down / srk_proj_up / srk_proj_wgrad,
through ops.ProjFn) against torch's float64 conv2d / conv_transpose2d on the SAME 16-bit-rounded operands.
Reference: models/ddbpn.py:10-24 (`projection_conv`: kernel 8, stride 4, padding 2) and :42-62 (how DenseProjection chains them)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    import sr_amd
    return sr_amd


def _ref(x_nhwc, w, b, up):
    x = x_nhwc.permute(0, 3, 1, 2).double()
    f = torch.nn.functional.conv_transpose2d if up else torch.nn.functional.conv2d
    return f(x, w.double(), None if b is None else b.double(), stride=4, padding=2)


# LR-side dims: tile multiples, ragged in both directions, smaller than one tile, several images, the reference's batch shape
@pytest.mark.parametrize("n,lh,lw", [(1, 4, 8), (2, 5, 11), (1, 1, 1), (3, 12, 7), (2, 48, 48)])
@pytest.mark.parametrize("up", [True, False])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("bias", [True, False])
def test_projection_forward_and_gradients(A, n, lh, lw, up, dt, bias):
    from sr_amd import ops
    g = torch.Generator().manual_seed(1000 * n + 10 * lh + lw + int(up))
    h, wd = (lh, lw) if up else (4 * lh, 4 * lw)
    x = torch.randn(n, h, wd, 32, generator=g).to(dt)
    # the kernels round the fp32 weights to the storage type: the reference takes the rounded values
    w = (torch.randn(32, 32, 8, 8, generator=g) * 0.05).to(dt).float()
    b = torch.randn(32, generator=g) if bias else None
    xd = x.cuda().requires_grad_(True)
    wdv = w.cuda().requires_grad_(True)
    bd = b.cuda().requires_grad_(True) if bias else None
    assert ops.proj_ok(xd, wdv, 4, 2, up)
    y = (ops.conv_transpose_general if up else ops.conv_general)(xd, wdv, bd, stride=4, pad=2)
    xr = x.double().requires_grad_(True)
    wr = w.double().requires_grad_(True)
    br = b.double().requires_grad_(True) if bias else None
    yr = _ref(xr, wr, br, up).permute(0, 2, 3, 1)
    assert tuple(y.shape) == tuple(yr.shape)
    eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    err = (y.detach().cpu().double() - yr.detach()).abs().max().item()
    scale = yr.detach().abs().max().item()
    assert err <= 1.5 * eps * scale + 1e-6, f"forward: {err} vs scale {scale}"
    # gradients: a 16-bit upstream gradient, the same one on both sides
    gy = torch.randn(yr.shape, generator=g).to(dt)
    y.backward(gy.cuda())
    yr.backward(gy.double())
    gx = xd.grad.cpu().double()
    assert (gx - xr.grad).abs().max().item() <= 1.5 * eps * xr.grad.abs().max().item() + 1e-6
    # fp32 sums over all pixels of products of 16-bit values: fp32 accumulation error only
    gw = wdv.grad.cpu().double()
    assert (gw - wr.grad).abs().max().item() <= 2e-4 * wr.grad.abs().max().item() + 1e-5
    if bias:
        assert (bd.grad.cpu().double() - br.grad).abs().max().item() <= 2e-4 * br.grad.abs().max().item() + 1e-4


def test_weight_gradient_accumulates_into_an_existing_grad(A):
    from sr_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 6, 9, 32, generator=g).to(torch.bfloat16).cuda().requires_grad_(True)
    w = (torch.randn(32, 32, 8, 8, generator=g) * 0.05).cuda().requires_grad_(True)
    b = torch.randn(32, generator=g).cuda().requires_grad_(True)
    y = ops.conv_transpose_general(x, w, b, stride=4, pad=2)
    gy = torch.randn(y.shape, generator=g).to(torch.bfloat16).cuda()
    y.backward(gy)
    first, firstb = w.grad.clone(), b.grad.clone()
    ptr, ptrb = w.grad.data_ptr(), b.grad.data_ptr()
    y2 = ops.conv_transpose_general(x, w, b, stride=4, pad=2)
    y2.backward(gy)
    assert w.grad.data_ptr() == ptr and b.grad.data_ptr() == ptrb
    torch.testing.assert_close(w.grad, 2 * first, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(b.grad, 2 * firstb, rtol=1e-6, atol=1e-5)


def test_shapes_outside_the_direct_kernels_keep_the_column_path(A):
    """Scale 2 / 8 projections, fp32 storage, other channel counts: proj_ok says no and the im2col / col2im path answers."""
    from sr_amd import ops
    x = torch.randn(1, 8, 8, 32).cuda()
    w = torch.randn(32, 32, 8, 8).cuda()
    assert not ops.proj_ok(x, w, 4, 2, True)                                    # fp32
    xb = x.to(torch.bfloat16)
    assert not ops.proj_ok(xb, torch.randn(32, 32, 6, 6).cuda(), 2, 2, True)    # scale 2
    assert not ops.proj_ok(xb[:, :7], w, 4, 2, False)                           # HR height not a multiple of 4
    assert ops.proj_ok(xb, w, 4, 2, False)


def test_ddbpn_training_step_uses_the_direct_projections(A, monkeypatch):
    """A D-DBPN x4 training step in bf16 issues 33 forward projections and none of the im2col launches."""
    from sr_amd import ops, _lib as L
    calls = {}
    real = L.call

    def spy(name, args, stream):
        calls[name] = calls.get(name, 0) + 1
        return real(name, args, stream)
    monkeypatch.setattr(L, "call", spy)
    torch.manual_seed(0)
    m = A.DDBPN(scale_factor=4, precision="bf16").cuda()
    lr = torch.rand(2, 3, 16, 16).cuda()
    hr = torch.rand(2, 3, 64, 64).cuda()
    loss = (m(lr) - hr).abs().mean()
    loss.backward()
    ops.flush_wgrads()
    assert calls.get("srk_proj_up", 0) + calls.get("srk_proj_down", 0) == 66, calls      # 33 forward + 33 data gradients
    assert calls.get("srk_proj_wgrad", 0) == 33
    assert calls.get("srk_unfold_nhwc", 0) == 0 and calls.get("srk_fold_nhwc", 0) == 0
    for p in m.parameters():
        if p.requires_grad:
            assert p.grad is not None and torch.isfinite(p.grad).all()


@pytest.mark.parametrize("up", [True, False])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("nslope", [1, 32])
def test_fused_prelu_matches_the_two_launch_form(A, up, dt, nslope):
    """ops.proj_prelu (the PReLU in the projection's epilogue, ddbpn.py:42-53) against ops.prelu behind the plain projection: the
    activation is applied to the STORED conv output, so the forward is bit-identical; all four gradients follow."""
    from sr_amd import ops
    g = torch.Generator().manual_seed(77 + int(up) + nslope)
    n, lh, lw = 2, 7, 10
    h, wd = (lh, lw) if up else (4 * lh, 4 * lw)
    x0 = torch.randn(n, h, wd, 32, generator=g).to(dt).cuda()
    w0 = (torch.randn(32, 32, 8, 8, generator=g) * 0.05).cuda()
    b0 = torch.randn(32, generator=g).cuda()
    s0 = (torch.rand(nslope, generator=g) * 0.5 - 0.1).cuda()            # a few negative slopes too
    outs = []
    for fused in (True, False):
        x, w, b, s = (t.clone().requires_grad_(True) for t in (x0, w0, b0, s0))
        if fused:
            y = ops.proj_prelu(x, w, b, s, up=up)
        else:
            y = ops.prelu((ops.conv_transpose_general if up else ops.conv_general)(x, w, b, stride=4, pad=2), s)
        gy = torch.randn(y.shape, generator=torch.Generator().manual_seed(3)).to(dt).cuda()
        y.backward(gy)
        outs.append((y.detach(), x.grad, w.grad, b.grad, s.grad))
    f, u = outs
    assert torch.equal(f[0], u[0])
    assert torch.equal(f[1], u[1])
    for i in (2, 3, 4):
        torch.testing.assert_close(f[i], u[i], rtol=1e-5, atol=1e-5)


def test_grouped_projection_pack_follows_the_weights(A):
    """From the second forward on, D-DBPN's 33 projection weights are packed by ONE launch at the top of the forward (PackGroup):
    after an in-place parameter update the next forward must see the new weights -- bit-identical to a fresh model that packs per use --
    and a backward that runs after a LATER forward window packs the weights again instead of trusting the group's buffer."""
    import copy
    from sr_amd import ops
    torch.manual_seed(0)
    m = A.DDBPN(scale_factor=4, precision="bf16").cuda()
    lr = torch.rand(1, 3, 12, 12).cuda()
    y1 = m(lr)
    assert len(m._pack_group().proj_entries) == 33
    with torch.no_grad():
        for p in m.parameters():
            if p.requires_grad:
                p.mul_(1.01)
    y2 = m(lr)
    m2 = A.DDBPN(scale_factor=4, precision="bf16").cuda()
    m2.load_state_dict(copy.deepcopy(m.state_dict()))
    y3 = m2(lr)
    assert torch.equal(y2, y3) and not torch.equal(y1, y2)
    # a backward that runs after a LATER forward window of the same model: its token is stale, the weights are packed again there
    ga = torch.autograd.grad(y3.sum(), [p for p in m2.parameters() if p.requires_grad], allow_unused=True)
    y4 = m(lr)
    y5 = m(lr)
    gb = torch.autograd.grad(y4.sum(), [p for p in m.parameters() if p.requires_grad], allow_unused=True)
    assert torch.equal(y4, y5)
    for a_, b_ in zip(ga, gb):
        if a_ is not None:
            torch.testing.assert_close(a_, b_, rtol=1e-5, atol=1e-6)


def _ddbpn_grads(A, lr, hr):
    torch.manual_seed(0)
    m = A.DDBPN(scale_factor=4, precision="bf16").cuda()
    loss = (m(lr) - hr).abs().mean()
    loss.backward()
    return float(loss.detach()), {k: p.grad.detach().float().cpu() for k, p in m.named_parameters() if p.grad is not None}


def test_ddbpn_direct_path_agrees_with_the_column_path(A, monkeypatch):
    """The whole D-DBPN x4 training step in bf16 on the direct projection kernels + the shared gradient buffer of the concatenations,
    against (a) the im2col / col2im projections of round 2 (pinned by the `ddbpn_*` goldens) and (b) autograd's own sums of the
    gradient slices: same loss, every parameter gradient in the same direction and size (16-bit storage: rounding differs)."""
    from sr_amd import ops
    g = torch.Generator().manual_seed(11)
    lr, hr = torch.rand(2, 3, 16, 16, generator=g).cuda(), torch.rand(2, 3, 64, 64, generator=g).cuda()
    l0, g0 = _ddbpn_grads(A, lr, hr)
    monkeypatch.setattr(ops, "_SLICE_GACC", False)
    l1, g1 = _ddbpn_grads(A, lr, hr)
    monkeypatch.setattr(A.ops_proj, "_PROJ_OFF", True)      # (the knob lives with the projection ops since round 6)
    l2, g2 = _ddbpn_grads(A, lr, hr)
    assert abs(l0 - l1) < 1e-6 and abs(l0 - l2) < 2e-3 * abs(l2)
    assert set(g0) == set(g1) == set(g2)
    for name, other, lim in (("autograd sums", g1, 0.9995), ("column path", g2, 0.995)):
        worst = 1.0
        for k in g0:
            a_, b_ = g0[k].flatten().double(), other[k].flatten().double()
            if b_.norm() < 1e-9:
                continue
            cos = float(a_ @ b_ / (a_.norm() * b_.norm() + 1e-30))
            worst = min(worst, cos)
            assert cos > lim and 0.97 < float(a_.norm() / b_.norm()) < 1.03, f"{name}: {k}: cos {cos:.5f}, norm ratio {float(a_.norm() / b_.norm()):.4f}"
        print(f"D-DBPN bf16 gradients vs {name}: worst cosine {worst:.6f}")
