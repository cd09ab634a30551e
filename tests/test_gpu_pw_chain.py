"""GPU parity of the fused WDSR _Block_B kernels (csrc/pw_chain.hip: srk_pw_pack / srk_pw_forward / srk_pw_backward, through the
C ABI) against float64 references of the reference's block (models/wdsr.py:30-51):
    out = conv3x3(conv1x1(relu(conv1x1(x; F -> 6F)); 6F -> int(.8F)); -> F) * res_scale + x
The hidden tensor is rounded to the storage dtype in both (the HIP path keeps it in 16-bit MFMA operand registers)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {torch.float16: 6e-3, torch.bfloat16: 4e-2}
L2TOL = {torch.float16: 3e-2, torch.bfloat16: 8e-2}
DTYPES = [torch.bfloat16, torch.float16]


@pytest.fixture(scope="module")
def A():
    import sr_amd
    assert torch.cuda.is_available()
    sr_amd._lib.load()
    return sr_amd


def q(t, dt):
    return t.to(dt).double()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def relerr(got, ref):
    ref = ref.double()
    return float((got.double().cpu() - ref).abs().max() / max(1e-9, float(ref.abs().max())))


def l2err(got, ref):
    got, ref = got.double().cpu(), ref.double()
    return float((got - ref).norm() / max(1e-12, float(ref.norm())))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("f,tiles_per_cu,extra", [(128, 1, 37), (128, 2, 300), (64, 2, 5)])
def test_pw_forward_persistent_tiles(A, dt, f, tiles_per_cu, extra):
    """srk_pw_forward beyond one 256-pixel tile per CU takes the persistent kernel (csrc/pw_chain.hip, pw_fwd2_kernel): workgroups
    with 2 and 3 tiles (the weight ring runs across the tile switch, the next tile's pixels arrive under the current tile's last
    slices, results leave through LDS) and a ragged last tile, against float64 (models/wdsr.py:30-51: the block's two 1x1 convs)."""
    ops = A.ops
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    npix = 256 * cus * tiles_per_cu + extra
    chid, cmid = 6 * f, int(0.8 * f)
    cz = ops.pad16(cmid)
    x = rnd(1, 1, npix, f, seed=11)
    w1, b1 = rnd(chid, f, 1, 1, seed=12, scale=1.0 / np.sqrt(f)), rnd(chid, seed=13, scale=0.1)
    w2, b2 = rnd(cmid, chid, 1, 1, seed=14, scale=1.0 / np.sqrt(chid)), rnd(cmid, seed=15, scale=0.1)
    xr = q(x, dt).view(npix, f)
    hr = torch.relu(xr @ q(w1, dt).view(chid, f).t() + b1.double()).to(dt).double()
    zr = hr @ q(w2, dt).view(cmid, chid).t() + b2.double()
    pk = ops.pw_pack(torch.nn.Parameter(w1.cuda()), b1.cuda(), torch.nn.Parameter(w2.cuda()), b2.cuda(), dt)
    z = torch.full((1, 1, npix, cz), 7.0, dtype=dt, device="cuda")
    ops.pw_forward_raw(x.to(dt).cuda(), pk, z)
    torch.cuda.synchronize()
    got = z.view(npix, cz).double().cpu()
    worst = float((got[:, :cmid] - zr).abs().max())
    assert relerr(got[:, :cmid], zr) < TOL[dt], f"max abs err {worst}"
    for lo in range(0, npix, 256 * cus):                          # no tile of any round is off (a single bad tile hides in the norm)
        hi = min(npix, lo + 256 * cus)
        assert relerr(got[lo:hi, :cmid], zr[lo:hi]) < TOL[dt], f"pixels {lo}..{hi}"
    assert relerr(got[-512:, :cmid], zr[-512:]) < TOL[dt], "the ragged tail"
    assert float(got[:, cmid:].abs().max()) == 0.0, "padding channels are zeros"


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("f,npix", [(128, 256), (128, 1000), (128, 37), (64, 512), (64, 300)])
def test_pw_forward_and_backward_raw(A, dt, f, npix):
    """srk_pw_forward / srk_pw_backward on a flat pixel list (ragged sizes: the last workgroup / wave is partial), incl. the
    h / gh tensors the backward leaves for the weight-gradient GEMMs, against float64."""
    ops = A.ops
    chid, cmid = 6 * f, int(0.8 * f)
    cz = ops.pad16(cmid)
    x = rnd(1, 1, npix, f, seed=1)
    w1, b1 = rnd(chid, f, 1, 1, seed=2, scale=1.0 / np.sqrt(f)), rnd(chid, seed=3, scale=0.1)
    w2, b2 = rnd(cmid, chid, 1, 1, seed=4, scale=1.0 / np.sqrt(chid)), rnd(cmid, seed=5, scale=0.1)
    gz = rnd(1, 1, npix, cz, seed=6)
    gz[..., cmid:] = 0
    res = rnd(1, 1, npix, f, seed=7)
    # float64 reference with operands rounded to dt, h rounded to dt where it is stored / fed to the second GEMM
    xr = q(x, dt).view(npix, f).requires_grad_(True)
    w1r, w2r = q(w1, dt).view(chid, f), q(w2, dt).view(cmid, chid)
    pre = xr @ w1r.t() + b1.double()
    hr = torch.relu(pre).to(dt).double()
    zr = hr @ w2r.t() + b2.double()
    gzr = q(gz, dt).view(npix, cz)[:, :cmid]
    ghr = ((gzr @ w2r) * (pre > 0)).to(dt).double()
    gxr = ghr @ w1r + q(res, dt).view(npix, f)

    xd = x.to(dt).cuda()
    pk = ops.pw_pack(torch.nn.Parameter(w1.cuda()), b1.cuda(), torch.nn.Parameter(w2.cuda()), b2.cuda(), dt)
    z = torch.full((1, 1, npix, cz), 7.0, dtype=dt, device="cuda")
    ops.pw_forward_raw(xd, pk, z)
    torch.cuda.synchronize()
    assert relerr(z.view(npix, cz)[:, :cmid], zr.detach()) < TOL[dt]
    assert float(z.view(npix, cz)[:, cmid:].abs().max()) == 0.0, "padding channels are zeros"
    gx = torch.empty_like(xd)
    hid = torch.empty((1, 1, npix, chid), dtype=dt, device="cuda")
    ghid = torch.empty_like(hid)
    ops.pw_backward_raw(xd, gz.to(dt).cuda(), pk, gx, res=res.to(dt).cuda(), h_out=hid, gh_out=ghid)
    gx2 = torch.empty_like(xd)
    ops.pw_backward_raw(xd, gz.to(dt).cuda(), pk, gx2, res=None)
    torch.cuda.synchronize()
    assert relerr(hid.view(npix, chid), hr.detach()) < TOL[dt]
    assert l2err(ghid.view(npix, chid), ghr.detach()) < L2TOL[dt]
    assert l2err(gx.view(npix, f), gxr.detach()) < L2TOL[dt]
    assert l2err(gx2.view(npix, f), (gxr - q(res, dt).view(npix, f)).detach()) < L2TOL[dt]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("f,n,h,w", [(128, 2, 12, 12), (128, 1, 17, 9), (64, 2, 16, 16)])
def test_wdsr_block_b_fused_vs_float64(A, dt, f, n, h, w):
    """ops.wdsr_block_b (the fused form) forward and every gradient against float64 autograd of the reference's block."""
    ops = A.ops
    tol = TOL[dt] * 2
    shapes = [(6 * f, f, 1), (int(0.8 * f), 6 * f, 1), (f, int(0.8 * f), 3)]
    scale = 0.5
    x = rnd(n, f, h, w, seed=1)
    gy = rnd(n, f, h, w, seed=9)
    ws = [rnd(co, ci, k, k, seed=10 + i, scale=1.0 / np.sqrt(ci * k * k)) for i, (co, ci, k) in enumerate(shapes)]
    bs = [rnd(co, seed=20 + i, scale=0.1) for i, (co, ci, k) in enumerate(shapes)]
    xr = q(x, dt).requires_grad_(True)
    wr = [q(w_, dt).requires_grad_(True) for w_ in ws]
    br = [b_.double().requires_grad_(True) for b_ in bs]
    a = F.relu(F.conv2d(xr, wr[0], br[0]))
    a = F.conv2d(a, wr[1], br[1])
    a = F.conv2d(a, wr[2], br[2], padding=1)
    yr = a * scale + xr
    yr.backward(q(gy, dt))
    xd = x.permute(0, 2, 3, 1).contiguous().to(dt).cuda().requires_grad_(True)
    wd = [torch.nn.Parameter(w_.cuda()) for w_ in ws]
    bd = [torch.nn.Parameter(b_.cuda()) for b_ in bs]
    assert ops.pw_ok(xd, wd[0], wd[1]), "this shape takes the fused kernels"
    y = ops.wdsr_block_b(xd, list(zip(wd, bd)), scale=scale)
    y.backward(gy.permute(0, 2, 3, 1).contiguous().to(dt).cuda())
    torch.cuda.synchronize()
    assert relerr(y.detach().permute(0, 3, 1, 2), yr.detach()) < tol
    assert l2err(xd.grad.permute(0, 3, 1, 2), xr.grad) < L2TOL[dt]
    for i in range(3):
        assert l2err(wd[i].grad, wr[i].grad) < L2TOL[dt], f"dw{i}"
        assert l2err(bd[i].grad, br[i].grad) < L2TOL[dt], f"db{i}"


def test_fused_block_equals_the_three_launch_chain(A):
    """Same arithmetic as the unfused conv chain (same operand rounding, fp32 accumulation; only the K order of the second GEMM
    differs): outputs within 2 ulp of the storage dtype's rounding, the ReLU mask of the backward identical."""
    ops = A.ops
    dt = torch.bfloat16
    f, n, h, w = 128, 2, 24, 24
    shapes = [(6 * f, f, 1), (int(0.8 * f), 6 * f, 1), (f, int(0.8 * f), 3)]
    x = rnd(n, h, w, f, seed=1).to(dt).cuda()
    wd = [torch.nn.Parameter(rnd(co, ci, k, k, seed=10 + i, scale=1.0 / np.sqrt(ci * k * k)).cuda()) for i, (co, ci, k) in enumerate(shapes)]
    bd = [torch.nn.Parameter(rnd(co, seed=20 + i, scale=0.1).cuda()) for i, (co, ci, k) in enumerate(shapes)]
    g = rnd(n, h, w, f, seed=3).to(dt).cuda()
    outs = []
    for fused in (True, False):
        xi = x.clone().requires_grad_(True)
        for p_ in wd + bd:
            p_.grad = None
        if fused:
            y = ops.wdsr_block_b(xi, list(zip(wd, bd)), scale=1.0)
        else:
            y = ops.conv_chain(xi, list(zip(wd, bd)), [True, False, False], scale=1.0)
        y.backward(g)
        torch.cuda.synchronize()
        outs.append((y.detach().float(), xi.grad.float(), [p_.grad.clone() for p_ in wd + bd]))
    (yf, gxf, gpf), (yc, gxc, gpc) = outs
    assert float((yf - yc).abs().max()) <= 2 ** -6 * float(yc.abs().max())
    assert l2err(gxf, gxc.cpu()) < 1e-2
    for a_, b_ in zip(gpf, gpc):
        assert l2err(a_, b_.cpu()) < 1e-2


def test_weight_norm_group_matches_torch(A):
    """ops.WeightNormGroup (csrc/wn.hip: every weight-normed conv of a model in one launch per direction) against
    torch._weight_norm and its autograd backward (nn.utils.weight_norm, models/wdsr.py:62)."""
    import torch.nn as nn
    torch.manual_seed(3)
    convs = [nn.utils.weight_norm(nn.Conv2d(ci, co, k, padding=k // 2)).cuda() for ci, co, k in [(3, 48, 5), (128, 768, 1), (768, 102, 1), (102, 128, 3), (7, 5, 3)]]
    with torch.no_grad():
        for c in convs:
            c.weight_g.mul_(torch.rand_like(c.weight_g) + 0.5)
    grp = A.ops.WeightNormGroup(convs)
    ws = grp.weights()
    gens = [torch.randn_like(w) for w in ws]
    torch.autograd.backward(ws, gens)
    got = [(w.detach().clone(), c.weight_v.grad.clone(), c.weight_g.grad.clone()) for w, c in zip(ws, convs)]
    for c in convs:
        c.weight_v.grad = c.weight_g.grad = None
    for (w, gv, gg), c, gen in zip(got, convs, gens):
        ref = torch._weight_norm(c.weight_v, c.weight_g, 0)
        ref.backward(gen)
        assert float((w - ref.detach()).abs().max()) <= 2e-6 * float(ref.abs().max())
        assert float((gv - c.weight_v.grad).abs().max()) <= 1e-5 * float(c.weight_v.grad.abs().max())
        assert float((gg - c.weight_g.grad).abs().max()) <= 1e-5 * float(c.weight_g.grad.abs().max()) + 1e-6
    p0 = ws[0].data_ptr()
    assert grp.weights()[0].data_ptr() == p0, "the effective weights keep their address from step to step"


def test_wdsr_step_uses_grouped_parameter_launches(A):
    """A WDSR-B training step issues ONE weight-norm launch per direction and serves the packed weights of its 51 convs from the
    model's PackGroup (was: ~100 torch weight-norm launches + ~100 per-use packs), and two steps give the same result as the
    per-conv torch path (SRK_NO_PW-independent: compared through the parameters' gradients)."""
    torch.manual_seed(0)
    m = A.WDSR(type="B", n_feats=128, n_resblocks=2, scale_factor=2, precision="bf16").cuda()
    x, hr = torch.rand(2, 3, 24, 24).cuda(), torch.rand(2, 3, 48, 48).cuda()
    for _ in range(2):
        for p in m.parameters():
            p.grad = None
        m.training_step({"lr": x, "hr": hr}, 0)["loss"].backward()
    torch.cuda.synchronize()
    grp = m._pack_group()
    assert len(grp.pw_entries) == 2 and len(grp.entries) >= 2 * 2 + 3, (len(grp.pw_entries), len(grp.entries))
    got = {k: p.grad.clone() for k, p in m.named_parameters()}
    # reference: the same model with torch's weight norm per conv and per-use packs
    from sr_amd.models import wdsr as W
    for p in m.parameters():
        p.grad = None
    y = None
    with A.ops.forward_scope(None):
        mean = m.rgb_mean.view(3).contiguous()
        s = A.ops.skip_conv(x, W._wn_weight(m.skip[0]), m.skip[0].bias, mean, 2, m.compute_dtype)
        f = A.ops.head_conv(x, W._wn_weight(m.head[0]), m.head[0].bias, mean, m.compute_dtype)
        for blk in m.body:
            f = blk.nhwc(f)
        y = A.ops.tail_conv(f, W._wn_weight(m.tail[0]), m.tail[0].bias, res=s, post_add=mean, ps_r=2)
    m._calculate_losses(img_sr=y, img_hr=hr)["loss"].backward()
    torch.cuda.synchronize()
    for k, p in m.named_parameters():
        ref = p.grad
        assert l2err(got[k], ref.cpu()) < 2e-3, k


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("f,npix", [(128, 128), (128, 5000), (128, 37), (64, 700), (64, 36864)])
def test_pw_wgrad_raw(A, dt, f, npix):
    """srk_pw_wgrad (h and gh re-computed per 128-pixel tile, never in HBM) against float64: dW1, db1, dW2 incl. ragged last tiles and
    pixel ranges of unequal length."""
    ops = A.ops
    chid, cmid = 6 * f, int(0.8 * f)
    cz = ops.pad16(cmid)
    x = rnd(1, 1, npix, f, seed=1)
    w1, b1 = rnd(chid, f, 1, 1, seed=2, scale=1.0 / np.sqrt(f)), rnd(chid, seed=3, scale=0.1)
    w2, b2 = rnd(cmid, chid, 1, 1, seed=4, scale=1.0 / np.sqrt(chid)), rnd(cmid, seed=5, scale=0.1)
    gz = rnd(1, 1, npix, cz, seed=6)
    gz[..., cmid:] = 0
    xr = q(x, dt).view(npix, f)
    w1r, w2r = q(w1, dt).view(chid, f), q(w2, dt).view(cmid, chid)
    pre = xr @ w1r.t() + b1.double()
    hr = torch.relu(pre).to(dt).double()
    gzr = q(gz, dt).view(npix, cz)[:, :cmid]
    ghr = ((gzr @ w2r) * (pre > 0)).to(dt).double()
    dw1r, db1r, dw2r, db2r = ghr.t() @ xr, ghr.sum(0), gzr.t() @ hr, gzr.sum(0)
    pk = ops.pw_pack(torch.nn.Parameter(w1.cuda()), b1.cuda(), torch.nn.Parameter(w2.cuda()), b2.cuda(), dt)
    dw1, db1, dw2, db2 = ops.pw_wgrad_raw(x.to(dt).cuda(), gz.to(dt).cuda(), pk, (chid, f, 1, 1), (cmid, chid, 1, 1))
    torch.cuda.synchronize()
    assert l2err(dw1.view(chid, f), dw1r) < L2TOL[dt] / 4
    assert l2err(db1, db1r) < L2TOL[dt] / 4
    assert l2err(dw2.view(cmid, chid), dw2r) < L2TOL[dt] / 4
    assert l2err(db2, db2r) < 1e-3
    # per ELEMENT: the reference above is built from the ROUNDED operands incl. the 16-bit-rounded h and gh, so what
    # is left is accumulation order plus the few h / gh elements whose fp32 pre-activation sits on a rounding boundary -- a fragment
    # permutation slip that moved one (row, column) pair would be an O(1) error in that element, far above these bounds
    assert relerr(dw1.view(chid, f), dw1r) < 6e-3
    assert relerr(dw2.view(cmid, chid), dw2r) < 6e-3
    assert relerr(db1, db1r) < 6e-3
