"""GPU parity of the HIP ops (through the C ABI) against float64 CPU references built from
oracle.functional / torch.nn.functional -- forward values and every gradient.

Tolerances are relative to the reference's max magnitude.  fp32 (exact-product MFMA, fp32 accumulate)
must meet north_star's "within 1e-3" with a wide margin (2e-4).  bf16/fp16 store activations in 16 bits:
the reference is computed in float64 from inputs/weights rounded to that dtype, so what remains is the
rounding of stored intermediates (2^-9 / 2^-11 per stored tensor).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 2e-4, torch.float16: 6e-3, torch.bfloat16: 4e-2}
DTYPES = [torch.float32, torch.bfloat16, torch.float16]


@pytest.fixture(scope="module")
def A():
    import sr_amd
    assert torch.cuda.is_available()
    sr_amd._lib.load()
    return sr_amd


def q(t, dt):
    """round to the compute dtype, return float64 on CPU"""
    return t.to(dt).double()


def nhwc(x, dt, dev="cuda"):
    """NCHW float -> NHWC dt with channels zero-padded to 16 (host-side helper, torch ops only)."""
    n, c, h, w = x.shape
    cp = (c + 15) // 16 * 16
    o = torch.zeros((n, h, w, cp), dtype=dt)
    o[..., :c] = x.permute(0, 2, 3, 1).to(dt)
    return o.to(dev)


def nchw(t, c):
    return t[..., :c].permute(0, 3, 1, 2).double().cpu()


def relerr(got, ref):
    ref = ref.double()
    return float((got.double().cpu() - ref).abs().max() / max(1e-9, float(ref.abs().max())))


def check(name, got, ref, tol):
    e = relerr(got, ref)
    assert np.isfinite(e) and e < tol, f"{name}: rel err {e:.3e} >= {tol:.1e}"
    return e


def check_l2(name, got, ref, tol):
    """For gradients that pass through a ReLU mask: a pre-activation within rounding distance of zero can
    take the other branch than in the float64 reference and changes a handful of elements by O(1), so the
    criterion is the relative L2 error plus a bound on the fraction of outliers (not the max error)."""
    got, ref = got.double().cpu(), ref.double()
    l2 = float((got - ref).norm() / max(1e-12, float(ref.norm())))
    frac = float(((got - ref).abs() > 10 * tol * ref.abs().max()).double().mean())
    assert np.isfinite(l2) and l2 < tol and frac < 2e-3, f"{name}: rel L2 err {l2:.3e} (tol {tol:.1e}), outliers {frac:.2e}"
    return l2


# fp32: ONE flipped mask element in a 2x64x48x48 activation moves a 64x64x3x3 weight gradient by ~4e-3 in rel. L2
L2TOL = {torch.float32: 1e-2, torch.float16: 3e-2, torch.bfloat16: 8e-2}


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("n,ci,co,h,w,k", [(2, 64, 64, 48, 48, 3), (1, 16, 16, 7, 9, 3), (1, 16, 48, 17, 33, 3),
                                           (1, 256, 256, 20, 18, 3), (1, 128, 768, 10, 11, 1), (1, 768, 102, 9, 10, 1),
                                           (1, 102, 128, 10, 9, 3), (3, 64, 32, 5, 5, 1), (1, 576, 64, 12, 12, 3)])
def test_conv_fwd_bwd(A, dt, n, ci, co, h, w, k):
    """ConvFn: y = conv(x)*s + res ; grads wrt x, w, b, res."""
    tol = TOL[dt]
    x = rnd(n, ci, h, w, seed=1)
    wt = rnd(co, ci, k, k, seed=2, scale=1.0 / np.sqrt(ci * k * k))
    b = rnd(co, seed=3, scale=0.1)
    res = rnd(n, co, h, w, seed=4)
    gy = rnd(n, co, h, w, seed=5)
    s = 0.7
    # reference (float64, inputs rounded to dt; weights are rounded by the pack kernel)
    xr, rr = q(x, dt).requires_grad_(True), q(res, dt).requires_grad_(True)
    wr, br = q(wt, dt).requires_grad_(True), b.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, padding=k // 2) * s + rr
    yr.backward(q(gy, dt))
    # HIP
    xd = nhwc(x, dt).requires_grad_(True)
    rd = nhwc(res, dt).requires_grad_(True)
    wd = torch.nn.Parameter(wt.cuda())
    bd = torch.nn.Parameter(b.cuda())
    y = A.ops.conv(xd, wd, bd, res=rd, scale=s)
    y.backward(nhwc(gy, dt))
    torch.cuda.synchronize()
    check("y", nchw(y.detach(), co), yr.detach(), tol)
    cp = y.shape[3]
    assert float(y.detach()[..., co:].abs().max()) == 0.0 if cp > co else True, "padding channels must stay zero"
    check("dx", nchw(xd.grad, ci), xr.grad, tol)
    check("dres", nchw(rd.grad, co), rr.grad, tol)
    check("dw", wd.grad, wr.grad, tol)
    check("db", bd.grad, br.grad, tol)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("n,f,h,w,r", [(2, 64, 12, 12, 2), (1, 16, 5, 7, 3), (1, 16, 9, 6, 2), (1, 64, 6, 6, 4)])
def test_conv_pixelshuffle(A, dt, n, f, h, w, r):
    """conv F -> F r^2 with the PixelShuffle fused into the store (models/common.py:112-139) + backward."""
    tol = TOL[dt]
    co = f * r * r
    x = rnd(n, f, h, w, seed=1)
    wt = rnd(co, f, 3, 3, seed=2, scale=1.0 / np.sqrt(f * 9))
    b = rnd(co, seed=3, scale=0.1)
    gy = rnd(n, f, h * r, w * r, seed=5)
    xr, wr, br = q(x, dt).requires_grad_(True), q(wt, dt).requires_grad_(True), b.double().requires_grad_(True)
    yr = F.pixel_shuffle(F.conv2d(xr, wr, br, padding=1), r)
    yr.backward(q(gy, dt))
    xd = nhwc(x, dt).requires_grad_(True)
    wd, bd = torch.nn.Parameter(wt.cuda()), torch.nn.Parameter(b.cuda())
    y = A.ops.conv(xd, wd, bd, ps_r=r)
    assert tuple(y.shape) == (n, h * r, w * r, f)
    y.backward(nhwc(gy, dt))
    torch.cuda.synchronize()
    check("y", nchw(y.detach(), f), yr.detach(), tol)
    check("dx", nchw(xd.grad, f), xr.grad, tol)
    check("dw", wd.grad, wr.grad, tol)
    check("db", bd.grad, br.grad, tol)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("kind,f,h,w", [("res", 64, 12, 12), ("res", 16, 7, 5), ("res", 256, 8, 8),
                                        ("wdsr_a", 32, 9, 9), ("wdsr_b", 128, 8, 8), ("wdsr_b", 16, 7, 9)])
def test_conv_chain(A, dt, kind, f, h, w):
    """ResBlock (common.py:74-109) / WDSR _Block_A/_Block_B (wdsr.py:9-51) incl. the 102-channel tensors."""
    tol = TOL[dt] * 2
    if kind == "res":
        shapes, relus, scale = [(f, f, 3), (f, f, 3)], [True, False], 0.1
    elif kind == "wdsr_a":
        shapes, relus, scale = [(4 * f, f, 3), (f, 4 * f, 3)], [True, False], 1.0
    else:
        shapes, relus, scale = [(6 * f, f, 1), (int(0.8 * f), 6 * f, 1), (f, int(0.8 * f), 3)], [True, False, False], 1.0
    n = 2
    x = rnd(n, f, h, w, seed=1)
    gy = rnd(n, f, h, w, seed=9)
    ws = [rnd(co, ci, k, k, seed=10 + i, scale=1.0 / np.sqrt(ci * k * k)) for i, (co, ci, k) in enumerate(shapes)]
    bs = [rnd(co, seed=20 + i, scale=0.1) for i, (co, ci, k) in enumerate(shapes)]
    # reference: stored intermediates are rounded to dt like the HIP path stores them
    xr = q(x, dt).requires_grad_(True)
    wr = [q(w_, dt).requires_grad_(True) for w_ in ws]
    br = [b_.double().requires_grad_(True) for b_ in bs]
    a = xr
    for i, (w_, b_) in enumerate(zip(wr, br)):
        a = F.conv2d(a, w_, b_, padding=w_.shape[2] // 2)
        if relus[i]:
            a = F.relu(a)
    yr = a * scale + xr
    yr.backward(q(gy, dt))
    xd = nhwc(x, dt).requires_grad_(True)
    wd = [torch.nn.Parameter(w_.cuda()) for w_ in ws]
    bd = [torch.nn.Parameter(b_.cuda()) for b_ in bs]
    y = A.ops.conv_chain(xd, list(zip(wd, bd)), relus, scale=scale)
    y.backward(nhwc(gy, dt))
    torch.cuda.synchronize()
    check("y", nchw(y.detach(), f), yr.detach(), tol)
    check_l2("dx", nchw(xd.grad, f), xr.grad, L2TOL[dt])
    for i in range(len(ws)):
        check_l2(f"dw{i}", wd[i].grad, wr[i].grad, L2TOL[dt])
        check_l2(f"db{i}", bd[i].grad, br[i].grad, L2TOL[dt])


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("c,red,h,w", [(64, 16, 10, 10), (16, 4, 9, 7), (64, 16, 48, 48)])
def test_rcab(A, dt, c, red, h, w):
    """RCAB = conv, ReLU, conv, CALayer, += x (rcan.py:10-55)."""
    from oracle import functional as OF
    tol = TOL[dt] * 2
    n = 2
    x = rnd(n, c, h, w, seed=1)
    gy = rnd(n, c, h, w, seed=2)
    shp = {"p.body.0.weight": (c, c, 3, 3), "p.body.0.bias": (c,), "p.body.2.weight": (c, c, 3, 3), "p.body.2.bias": (c,),
           "p.body.3.conv_du.0.weight": (c // red, c, 1, 1), "p.body.3.conv_du.0.bias": (c // red,),
           "p.body.3.conv_du.2.weight": (c, c // red, 1, 1), "p.body.3.conv_du.2.bias": (c,)}
    raw = {k: rnd(*s, seed=30 + i, scale=(1.0 / np.sqrt(np.prod(s[1:])) if len(s) > 1 else 0.1)) for i, (k, s) in enumerate(shp.items())}
    conv_keys = ("p.body.0.weight", "p.body.2.weight")
    sd = {k: (q(v, dt) if k in conv_keys else v.double()).requires_grad_(True) for k, v in raw.items()}
    xr = q(x, dt).requires_grad_(True)
    yr = OF.rcab(sd, "p", xr)
    yr.backward(q(gy, dt))
    P = {k: torch.nn.Parameter(v.cuda()) for k, v in raw.items()}
    xd = nhwc(x, dt).requires_grad_(True)
    y = A.ops.rcab(xd, P["p.body.0.weight"], P["p.body.0.bias"], P["p.body.2.weight"], P["p.body.2.bias"],
                   P["p.body.3.conv_du.0.weight"], P["p.body.3.conv_du.0.bias"], P["p.body.3.conv_du.2.weight"], P["p.body.3.conv_du.2.bias"])
    y.backward(nhwc(gy, dt))
    torch.cuda.synchronize()
    check("y", nchw(y.detach(), c), yr.detach(), tol)
    check_l2("dx", nchw(xd.grad, c), xr.grad, L2TOL[dt])
    for k in raw:
        check_l2(k, P[k].grad, sd[k].grad, L2TOL[dt])


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("g0,g,c,h,w", [(64, 64, 8, 8, 8), (16, 32, 3, 7, 9), (64, 32, 6, 17, 16)])
def test_rdb(A, dt, g0, g, c, h, w):
    """Residual dense block with slice writes instead of torch.cat (rdn.py:9-40)."""
    from oracle import functional as OF
    tol = TOL[dt] * 2
    n = 1
    x = rnd(n, g0, h, w, seed=1)
    gy = rnd(n, g0, h, w, seed=2)
    shp = {}
    for i in range(c):
        shp[f"p.convs.{i}.conv.0.weight"] = (g, g0 + i * g, 3, 3)
        shp[f"p.convs.{i}.conv.0.bias"] = (g,)
    shp["p.LFF.weight"] = (g0, g0 + c * g, 1, 1)
    shp["p.LFF.bias"] = (g0,)
    raw = {k: rnd(*s, seed=40 + i, scale=(1.0 / np.sqrt(np.prod(s[1:])) if len(s) > 1 else 0.1)) for i, (k, s) in enumerate(shp.items())}
    sd = {k: (q(v, dt) if k.endswith("weight") else v.double()).requires_grad_(True) for k, v in raw.items()}
    xr = q(x, dt).requires_grad_(True)
    yr = OF.rdb(sd, "p", xr, c)
    yr.backward(q(gy, dt))
    P = {k: torch.nn.Parameter(v.cuda()) for k, v in raw.items()}
    xd = nhwc(x, dt).requires_grad_(True)
    y = A.ops.rdb(xd, [(P[f"p.convs.{i}.conv.0.weight"], P[f"p.convs.{i}.conv.0.bias"]) for i in range(c)],
                  (P["p.LFF.weight"], P["p.LFF.bias"]))
    y.backward(nhwc(gy, dt))
    torch.cuda.synchronize()
    check("y", nchw(y.detach(), g0), yr.detach(), tol)
    check_l2("dx", nchw(xd.grad, g0), xr.grad, L2TOL[dt])
    for k in raw:
        check_l2(k, P[k].grad, sd[k].grad, L2TOL[dt])


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cin,f,k,h,w", [(3, 64, 3, 12, 12), (1, 16, 3, 7, 9), (3, 128, 3, 9, 8)])
def test_head_conv(A, dt, cin, f, k, h, w):
    """sub_mean + head conv from an NCHW fp32 image (boundary im2col + 1x1 MFMA conv)."""
    tol = TOL[dt]
    n = 2
    x = rnd(n, cin, h, w, seed=1) * 0.5 + 0.5
    wt = rnd(f, cin, k, k, seed=2, scale=1.0 / np.sqrt(cin * k * k))
    b = rnd(f, seed=3, scale=0.1)
    sub = torch.tensor([0.4488, 0.4371, 0.4040][:cin])
    gy = rnd(n, f, h, w, seed=4)
    wr, br = q(wt, dt).requires_grad_(True), b.double().requires_grad_(True)
    xin = q(x - sub.view(1, -1, 1, 1), dt)        # the unfold kernel subtracts in fp32, then rounds
    yr = F.conv2d(xin, wr, br, padding=k // 2)
    yr.backward(q(gy, dt))
    wd, bd = torch.nn.Parameter(wt.cuda()), torch.nn.Parameter(b.cuda())
    y = A.ops.head_conv(x.cuda(), wd, bd, sub.cuda(), dt)
    y.backward(nhwc(gy, dt))
    torch.cuda.synchronize()
    check("y", nchw(y.detach(), f), yr.detach(), tol)
    check("dw", wd.grad, wr.grad, tol)
    check("db", bd.grad, br.grad, tol)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("f,cout,r,h,w", [(64, 3, 1, 20, 20), (16, 1, 1, 7, 9), (128, 48, 4, 9, 9), (16, 12, 2, 6, 7), (16, 27, 3, 5, 5)])
def test_tail_and_skip_conv(A, dt, f, cout, r, h, w):
    """tail conv -> NCHW fp32 (+PixelShuffle, + skip branch, + mean) as EDSR/RCAN/RDN/WDSR end (edsr.py:49-52, wdsr.py:107-115)."""
    tol = TOL[dt]
    n = 2
    cimg = cout // (r * r)
    x = rnd(n, f, h, w, seed=1)
    img = rnd(n, 3, h, w, seed=6) * 0.5 + 0.5
    wt = rnd(cout, f, 3, 3, seed=2, scale=1.0 / np.sqrt(f * 9))
    b = rnd(cout, seed=3, scale=0.1)
    ws = rnd(cout, 3, 5, 5, seed=7, scale=1.0 / np.sqrt(75))
    bs_ = rnd(cout, seed=8, scale=0.1)
    mean = torch.tensor([0.4488, 0.4371, 0.4040, 0.5, 0.5])[:cimg].contiguous()
    sub = torch.tensor([0.4488, 0.4371, 0.4040])
    gy = rnd(n, cimg, h * r, w * r, seed=5)
    xr, wr, br = q(x, dt).requires_grad_(True), q(wt, dt).requires_grad_(True), b.double().requires_grad_(True)
    wsr, bsr = q(ws, dt).requires_grad_(True), bs_.double().requires_grad_(True)
    sr = F.conv2d(q(img - sub.view(1, 3, 1, 1), dt), wsr, bsr, padding=2)
    yr = F.conv2d(xr, wr, br, padding=1) + sr
    if r > 1:
        yr = F.pixel_shuffle(yr, r)
    yr = yr + mean.double().view(1, -1, 1, 1)
    yr.backward(gy.double())
    xd = nhwc(x, dt).requires_grad_(True)
    wd, bd = torch.nn.Parameter(wt.cuda()), torch.nn.Parameter(b.cuda())
    wsd, bsd = torch.nn.Parameter(ws.cuda()), torch.nn.Parameter(bs_.cuda())
    s = A.ops.skip_conv(img.cuda(), wsd, bsd, sub.cuda(), r, dt)
    y = A.ops.tail_conv(xd, wd, bd, res=s, post_add=mean.cuda(), ps_r=r)
    assert y.dtype == torch.float32 and tuple(y.shape) == (n, cimg, h * r, w * r)
    y.backward(gy.cuda())
    torch.cuda.synchronize()
    check("y", y.detach().cpu(), yr.detach(), tol)
    check("dx", nchw(xd.grad, f), xr.grad, tol)
    check("dw", wd.grad, wr.grad, tol)
    check("db", bd.grad, br.grad, tol)
    check("dw_skip", wsd.grad, wsr.grad, tol)
    check("db_skip", bsd.grad, bsr.grad, tol)


@pytest.mark.parametrize("dt", DTYPES)
def test_layout_roundtrip(A, dt):
    x = rnd(2, 20, 6, 9, seed=1)
    t = A.ops.to_nhwc(x.cuda(), dt)
    assert tuple(t.shape) == (2, 6, 9, 32) and float(t[..., 20:].abs().max()) == 0.0
    back = A.ops.to_nchw(t, 20)
    assert torch.equal(back.cpu(), x.to(dt).float())
    y = rnd(1, 3, 8, 12, seed=2)
    u = A.ops.to_nhwc(y.cuda(), dt, ps_r=2)
    assert torch.equal(A.ops.to_nchw(u, 12).cpu(), F.pixel_unshuffle(y, 2).to(dt).float())


def test_no_cpu_fallback(A):
    with pytest.raises(RuntimeError):
        A.ops.conv(torch.zeros(1, 4, 4, 16), torch.nn.Parameter(torch.zeros(16, 16, 3, 3)), None)


def test_bad_args_raise(A):
    w = torch.nn.Parameter(torch.zeros(16, 16, 5, 5).cuda())
    with pytest.raises(RuntimeError, match="not supported"):
        A.ops.conv(torch.zeros(1, 4, 4, 16, device="cuda"), w, None)


def test_tensors_beyond_2gib_are_chunked(A, monkeypatch):
    """Tensors of 2 GiB and more are processed in batch chunks (the persistent kernels use 32-bit buffer offsets).
    Exercised by lowering the limit instead of allocating gigabytes: results must equal the single-launch results."""
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(9)
    x = ((torch.rand(5, 20, 20, 64, generator=g) - 0.5)).to(dt).cuda()
    res = ((torch.rand(5, 20, 20, 64, generator=g) - 0.5)).to(dt).cuda()
    dy = ((torch.rand(5, 20, 20, 64, generator=g) - 0.5)).to(dt).cuda()
    w = torch.nn.Parameter(((torch.rand(64, 64, 3, 3, generator=g) - 0.5) * 0.1).cuda())
    b = torch.nn.Parameter(((torch.rand(64, generator=g) - 0.5) * 0.1).cuda())
    pk = A.ops.pack_conv(w, b, dt)

    def run():
        out = torch.empty_like(x)
        A.ops.conv_raw(x, pk, N=5, H=20, W=20, Cin=64, Cout=64, out=out, res=res, scale=0.5)
        dw, db = A.ops.wgrad_raw(x, dy, N=5, H=20, W=20, Cin=64, Cout=64, k=3, w_shape=(64, 64, 3, 3))
        torch.cuda.synchronize()
        return out, dw, db
    o1, w1, b1 = run()
    monkeypatch.setattr(A.ops, "_ADDR_LIMIT", 2 * 20 * 20 * 64 * 2 + 1)      # two images per chunk
    assert A.ops._batch_chunks(5, x) > 1
    o2, w2, b2 = run()
    assert torch.equal(o1, o2)
    assert float((w1 - w2).abs().max()) <= 1e-5 * float(w1.abs().max()) and float((b1 - b2).abs().max()) <= 1e-5 * float(b1.abs().max())
