"""GPU tests of conv_ks_kernel (csrc/conv_ks.hip): the 3x3 convolution for many input channels (EDSR-large's 256 -> 256
layers, models/edsr.py; RDN's dense layers, models/rdn.py:9-40) that srk_conv2d dispatches to for Cin = 64 k >= 128, Cout a
multiple of 64, 16-bit NHWC.  Every epilogue form (bias, ReLU, scale, residual, ReLU mask from a channel on, residual AND mask)
on aligned and ragged images, forward and data-gradient packs, against a float64 reference on the rounded operands."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    import sr_amd
    assert torch.cuda.is_available()
    sr_amd._lib.load()
    return sr_amd


def _rnd(g, *shape, scale=1.0):
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


CASES = [(16, 48, 48, 256, 256), (2, 20, 33, 128, 64), (1, 7, 9, 512, 128), (3, 16, 16, 192, 64), (1, 1, 1, 128, 64), (2, 17, 31, 256, 192)]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("form", ["bias_relu", "scale_res", "mask", "res_mask_from", "dgrad_mask"])
def test_conv_ks_against_float64(A, dt, form):
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(7)
    for (n, h, w, ci, co) in CASES:
        x = _rnd(g, n, h, w, ci).to(dt)
        wt = _rnd(g, co, ci, 3, 3, scale=1.0 / np.sqrt(9 * ci))
        b = _rnd(g, co, scale=0.2)
        res = _rnd(g, n, h, w, co).to(dt)
        mk = torch.relu(_rnd(g, n, h, w, co)).to(dt)
        dgrad = form == "dgrad_mask"
        wp, bp = torch.nn.Parameter(wt.to(dev)), torch.nn.Parameter(b.to(dev))
        kw = dict(relu=False, scale=1.0, res=None, mask=None, mask_from=0, use_bias=not dgrad)
        if dgrad:      # data gradient of a conv ci -> co' : input has co' = `ci` channels here, weight is [ci_out = ci][co] ...
            wt2 = _rnd(g, ci, co, 3, 3, scale=1.0 / np.sqrt(9 * ci))          # conv co -> ci ; its dgrad maps ci -> co
            wp = torch.nn.Parameter(wt2.to(dev))
            pk = A.ops.pack_conv(wp, None, dt, dgrad=True)
            kw.update(mask=mk.to(dev))
            wq = wt2.to(dt).double()
            ref = F.conv_transpose2d(x.double().permute(0, 3, 1, 2), wq, padding=1)       # [n, co, h, w]
            ref = torch.where(mk.double().permute(0, 3, 1, 2) > 0, ref, torch.zeros_like(ref))
        else:
            pk = A.ops.pack_conv(wp, bp, dt)
            wq = wt.to(dt).double()
            ref = F.conv2d(x.double().permute(0, 3, 1, 2), wq, b.double(), padding=1)
            if form == "bias_relu":
                kw.update(relu=True)
                ref = torch.relu(ref)
            elif form == "scale_res":
                kw.update(scale=0.1, res=res.to(dev))
                ref = ref * 0.1 + res.double().permute(0, 3, 1, 2)
            elif form == "mask":
                kw.update(mask=mk.to(dev))
                ref = torch.where(mk.double().permute(0, 3, 1, 2) > 0, ref, torch.zeros_like(ref))
            elif form == "res_mask_from":
                mf = 32 if co > 32 else 0
                kw.update(res=res.to(dev), mask=mk.to(dev), mask_from=mf)
                ref = ref + res.double().permute(0, 3, 1, 2)
                m = mk.double().permute(0, 3, 1, 2) > 0
                m[:, :mf] = True
                ref = torch.where(m, ref, torch.zeros_like(ref))
        out = torch.full((n, h, w, co), float("nan"), dtype=dt, device=dev)
        A.ops.conv_raw(x.to(dev), pk, N=n, H=h, W=w, Cin=ci, Cout=co, out=out, **kw)
        torch.cuda.synchronize()
        got = out.double().cpu().permute(0, 3, 1, 2)
        tol = (2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10) * max(1.0, float(ref.abs().max()))
        assert torch.isfinite(got).all(), (form, n, h, w, ci, co)
        assert float((got - ref).abs().max()) <= tol, (form, n, h, w, ci, co, float((got - ref).abs().max()), tol)


def test_conv_ks_on_channel_slices(A):
    """RDN's dense block: the conv reads the first 128 / 192 channels of a wider buffer and writes a 64-channel slice of it."""
    dev = torch.device("cuda")
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(3)
    n, h, w = 2, 19, 30
    feat = _rnd(g, n, h, w, 320).to(dt).to(dev)
    ref_feat = feat.clone()
    for cin in (128, 192):
        wt = _rnd(g, 64, cin, 3, 3, scale=1.0 / np.sqrt(9 * cin))
        b = _rnd(g, 64, scale=0.2)
        pk = A.ops.pack_conv(torch.nn.Parameter(wt.to(dev)), torch.nn.Parameter(b.to(dev)), dt)
        A.ops.conv_raw(feat[..., :cin], pk, N=n, H=h, W=w, Cin=cin, Cout=64, out=feat[..., cin:cin + 64], relu=True)
        torch.cuda.synchronize()
        r = torch.relu(F.conv2d(ref_feat[..., :cin].double().cpu().permute(0, 3, 1, 2), wt.to(dt).double(), b.double(), padding=1))
        got = feat[..., cin:cin + 64].double().cpu().permute(0, 3, 1, 2)
        assert float((got - r).abs().max()) <= 2.0 ** -7 * max(1.0, float(r.abs().max()))
        assert torch.equal(feat[..., cin + 64:], ref_feat[..., cin + 64:]) and torch.equal(feat[..., :cin], ref_feat[..., :cin])
        ref_feat = feat.clone()


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_conv_ks_pixelshuffle_store_and_shuffled_input(A, dt):
    """The upsampler of the 256-feature models (models/common.py:133: conv 256 -> 1024, PixelShuffle(2)): the fused
    pixel-shuffle store forward, and the data gradient that reads its incoming gradient through the same addressing."""
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(17)
    n, h, w, f, r = 2, 11, 13, 128, 2
    x = _rnd(g, n, f, h, w)
    wt = _rnd(g, f * r * r, f, 3, 3, scale=1.0 / np.sqrt(9 * f))
    b = _rnd(g, f * r * r, scale=0.1)
    xq = x.to(dt).double().requires_grad_(True)
    y = F.pixel_shuffle(F.conv2d(xq, wt.to(dt).double(), b.double(), padding=1), r)
    gy = _rnd(g, *y.shape).to(dt).double()
    y.backward(gy)
    xd = x.permute(0, 2, 3, 1).contiguous().to(dt).to(dev).requires_grad_(True)
    wp, bp = torch.nn.Parameter(wt.to(dev)), torch.nn.Parameter(b.to(dev))
    calls = []
    yd = A.ops.conv(xd, wp, bp, ps_r=r)
    yd.backward(gy.permute(0, 2, 3, 1).contiguous().to(dt).to(dev))
    torch.cuda.synchronize()
    tol = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
    got = yd.detach().double().cpu().permute(0, 3, 1, 2)
    assert float((got - y.detach()).abs().max()) <= tol * max(1.0, float(y.abs().max()))
    gx = xd.grad.double().cpu().permute(0, 3, 1, 2)
    assert float((gx - xq.grad).abs().max()) <= 2 * tol * max(1.0, float(xq.grad.abs().max()))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("form", ["bias_relu", "scale_res", "mask", "res_mask_from"])
def test_conv1x1_against_float64(A, dt, form):
    """conv1x1_kernel (csrc/conv1x1.hip): WDSR-B's pointwise convs (models/wdsr.py:30-51) and their data gradients."""
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(23)
    # (the last three: WDSR-B at its default width -- 128 -> 768, 768 -> 102 (stored as 112 channels), 102 -> 768 (input padded to 112))
    for (n, h, w, ci, co) in [(16, 48, 48, 64, 384), (16, 48, 48, 384, 64), (2, 9, 21, 128, 128), (1, 1, 1, 64, 64), (3, 17, 16, 320, 192),
                              (4, 24, 24, 128, 768), (4, 24, 24, 768, 102), (4, 24, 24, 102, 768)]:
        cip, cop = (ci + 15) // 16 * 16, (co + 15) // 16 * 16
        x = torch.zeros(n, h, w, cip, dtype=dt)
        x[..., :ci] = _rnd(g, n, h, w, ci).to(dt)
        wt = _rnd(g, co, ci, 1, 1, scale=1.0 / np.sqrt(ci))
        b = _rnd(g, co, scale=0.2)
        res = _rnd(g, n, h, w, cop).to(dt)
        mk = torch.relu(_rnd(g, n, h, w, cop)).to(dt)
        pk = A.ops.pack_conv(torch.nn.Parameter(wt.to(dev)), torch.nn.Parameter(b.to(dev)), dt)
        kw = dict(relu=False, scale=1.0, res=None, mask=None, mask_from=0)
        ref = F.conv2d(x[..., :ci].double().permute(0, 3, 1, 2), wt.to(dt).double(), b.double())
        resr, mkr = res[..., :co].double().permute(0, 3, 1, 2), mk[..., :co].double().permute(0, 3, 1, 2)
        if form == "bias_relu":
            kw.update(relu=True)
            ref = torch.relu(ref)
        elif form == "scale_res":
            kw.update(scale=0.3, res=res.to(dev))
            ref = ref * 0.3 + resr
        elif form == "mask":
            kw.update(mask=mk.to(dev))
            ref = torch.where(mkr > 0, ref, torch.zeros_like(ref))
        else:
            mf = 32
            kw.update(res=res.to(dev), mask=mk.to(dev), mask_from=mf)
            ref = ref + resr
            m = mkr > 0
            m[:, :mf] = True
            ref = torch.where(m, ref, torch.zeros_like(ref))
        out = torch.full((n, h, w, cop), float("nan"), dtype=dt, device=dev)
        A.ops.conv_raw(x.to(dev), pk, N=n, H=h, W=w, Cin=cip, Cout=cop, out=out, **kw)
        torch.cuda.synchronize()
        got = out[..., :co].double().cpu().permute(0, 3, 1, 2)
        assert torch.isfinite(out.float()).all(), "padding channels are written too"
        tol = (2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10) * max(1.0, float(ref.abs().max()))
        assert torch.isfinite(got).all() and float((got - ref).abs().max()) <= tol, (form, n, h, w, ci, co, float((got - ref).abs().max()), tol)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_conv_ks_partial_blocks_wdsr_shapes(A, dt):
    """WDSR-B's 3x3 conv 102 -> 128 (models/wdsr.py:30-51; the 102 channels live in 112-channel tensors) and its data gradient
    128 -> 102: a partial last 64-channel input block and stored channels that end inside the last 64-row output block."""
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(29)
    n, h, w, ci, co = 4, 24, 24, 102, 128
    wt = _rnd(g, co, ci, 3, 3, scale=1.0 / np.sqrt(9 * ci))
    b = _rnd(g, co, scale=0.2)
    wp, bp = torch.nn.Parameter(wt.to(dev)), torch.nn.Parameter(b.to(dev))
    tol = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
    # forward: x [.., 112] (last 10 channels zero), scale + residual
    x = torch.zeros(n, h, w, 112, dtype=dt)
    x[..., :ci] = _rnd(g, n, h, w, ci).to(dt)
    res = _rnd(g, n, h, w, co).to(dt)
    out = torch.full((n, h, w, co), float("nan"), dtype=dt, device=dev)
    A.ops.conv_raw(x.to(dev), A.ops.pack_conv(wp, bp, dt), N=n, H=h, W=w, Cin=112, Cout=co, out=out, scale=0.5, res=res.to(dev))
    ref = F.conv2d(x[..., :ci].double().permute(0, 3, 1, 2), wt.to(dt).double(), b.double(), padding=1) * 0.5 + res.double().permute(0, 3, 1, 2)
    torch.cuda.synchronize()
    assert float((out.double().cpu().permute(0, 3, 1, 2) - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))
    # data gradient: dy [.., 128] -> dx [.., 112] with a ReLU mask; the 10 padding channels of dx come out as zeros
    dy = _rnd(g, n, h, w, co).to(dt)
    mk = torch.relu(_rnd(g, n, h, w, 112)).to(dt)
    dx = torch.full((n, h, w, 112), float("nan"), dtype=dt, device=dev)
    A.ops.conv_raw(dy.to(dev), A.ops.pack_conv(wp, None, dt, dgrad=True), N=n, H=h, W=w, Cin=co, Cout=112, out=dx, mask=mk.to(dev), use_bias=False)
    refd = F.conv_transpose2d(dy.double().permute(0, 3, 1, 2), wt.to(dt).double(), padding=1)
    refd = torch.where(mk[..., :ci].double().permute(0, 3, 1, 2) > 0, refd, torch.zeros_like(refd))
    torch.cuda.synchronize()
    got = dx.double().cpu()
    assert float((got[..., :ci].permute(0, 3, 1, 2) - refd).abs().max()) <= tol * max(1.0, float(refd.abs().max()))
    assert float(got[..., ci:].abs().max()) == 0.0
