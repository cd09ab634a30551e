"""GPU tests of srk_conv_pair (two chained 3x3 64->64 convolutions in one launch, the small-batch form of ResBlock
models/common.py:74-109 and of RCAB's conv pair models/rcan.py:33-55, forward and backward).

The pair must give BIT-IDENTICAL results to the two srk_conv2d launches it replaces (same MFMA order, same epilogue
arithmetic, the intermediate rounded to the storage dtype in both), for every epilogue form the models use, on tile-aligned
and ragged images, and the block-level Functions must take it at the reference's batch and agree with the two-launch path
in outputs and every gradient."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    import sr_amd
    assert torch.cuda.is_available()
    sr_amd._lib.load()
    return sr_amd


def _rand(g, *shape, dt, dev, lo=-0.5):
    return (torch.rand(*shape, generator=g) + lo).to(dt).to(dev)


def _two_launch(A, x, pk1, pk2, *, relu_mid, scale_mid, mask, scale_out, res, use_bias):
    n, h, w, _ = x.shape
    mid = torch.empty_like(x)
    A.ops.conv_raw(x, pk1, N=n, H=h, W=w, Cin=64, Cout=64, out=mid, relu=relu_mid, scale=scale_mid, mask=mask, use_bias=use_bias)
    out = torch.empty_like(x)
    A.ops.conv_raw(mid, pk2, N=n, H=h, W=w, Cin=64, Cout=64, out=out, scale=scale_out, res=res, use_bias=use_bias)
    return mid, out


SHAPES = [(16, 48, 48), (1, 14, 14), (2, 28, 42), (3, 20, 33), (2, 5, 7), (1, 1, 1), (1, 15, 29), (5, 48, 48)]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("form", ["resblock", "resblock_bwd", "rcab", "rcab_bwd", "plain"])
def test_pair_is_bit_identical_to_two_launches(A, dt, form):
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(11)
    for (n, h, w) in SHAPES:
        x = _rand(g, n, h, w, 64, dt=dt, dev=dev)
        w1 = torch.nn.Parameter((torch.rand(64, 64, 3, 3, generator=g) - 0.5).mul(0.1).to(dev))
        w2 = torch.nn.Parameter((torch.rand(64, 64, 3, 3, generator=g) - 0.5).mul(0.1).to(dev))
        b1 = torch.nn.Parameter((torch.rand(64, generator=g) - 0.5).to(dev))
        b2 = torch.nn.Parameter((torch.rand(64, generator=g) - 0.5).to(dev))
        other = _rand(g, n, h, w, 64, dt=dt, dev=dev)
        maskt = _rand(g, n, h, w, 64, dt=dt, dev=dev)          # about half the entries <= 0
        maskt = torch.where(maskt > 0, maskt, torch.zeros_like(maskt))
        bwd = form.endswith("_bwd")
        pk1 = A.ops.pack_conv(w1, None if bwd else b1, dt, dgrad=bwd)
        pk2 = A.ops.pack_conv(w2, None if bwd else b2, dt, dgrad=bwd)
        kw = dict(relu_mid=False, scale_mid=1.0, mask=None, scale_out=1.0, res=None, use_bias=not bwd)
        if form == "resblock":
            kw.update(relu_mid=True, scale_out=0.1, res=x)
        elif form == "resblock_bwd":
            kw.update(scale_mid=0.1, mask=maskt, res=x)
        elif form == "rcab":
            kw.update(relu_mid=True)
        elif form == "rcab_bwd":
            kw.update(mask=maskt, res=other)
        mid_ref, out_ref = _two_launch(A, x, pk1, pk2, **kw)
        mid = torch.full_like(x, float("nan"))
        out = torch.full_like(x, float("nan"))
        A.ops.conv_pair_raw(x, pk1, pk2, out=out, mid=mid, **kw)
        torch.cuda.synchronize()
        assert torch.equal(mid.view(torch.int16), mid_ref.view(torch.int16)), (form, n, h, w, "mid")
        assert torch.equal(out.view(torch.int16), out_ref.view(torch.int16)), (form, n, h, w, "out")
        # without the intermediate store the output is the same
        out2 = torch.full_like(x, float("nan"))
        A.ops.conv_pair_raw(x, pk1, pk2, out=out2, **kw)
        torch.cuda.synchronize()
        assert torch.equal(out2.view(torch.int16), out_ref.view(torch.int16)), (form, n, h, w, "out, no mid")
        # channel-attention pooling of the output while it leaves: per-tile partial sums of out (* aux)
        T = A._lib.load().srk_conv_pair_tiles(1, h, w)
        for aux in (None, other):
            pool = torch.full((n, T, 64), float("nan"), dtype=torch.float32, device=dev)
            out3 = torch.empty_like(x)
            A.ops.conv_pair_raw(x, pk1, pk2, out=out3, pool=pool, pool_aux=aux, **kw)
            torch.cuda.synchronize()
            assert torch.equal(out3.view(torch.int16), out_ref.view(torch.int16))
            want = (out_ref.double() * (1.0 if aux is None else aux.double())).sum(dim=(1, 2))
            got = pool.double().sum(1)
            assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max()) + 1e-6, (form, n, h, w, aux is None)


def test_pair_on_channel_slices_of_wider_tensors(A):
    """pitch != 64: x, mid, out and res may be 64-channel slices of wider NHWC buffers."""
    dev = torch.device("cuda")
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(5)
    n, h, w = 2, 19, 30
    big = _rand(g, n, h, w, 192, dt=dt, dev=dev)
    x = big[..., 64:128]
    w1 = torch.nn.Parameter((torch.rand(64, 64, 3, 3, generator=g) - 0.5).mul(0.1).to(dev))
    w2 = torch.nn.Parameter((torch.rand(64, 64, 3, 3, generator=g) - 0.5).mul(0.1).to(dev))
    pk1, pk2 = A.ops.pack_conv(w1, None, dt), A.ops.pack_conv(w2, None, dt)
    obig = torch.zeros(n, h, w, 128, dtype=dt, device=dev)
    mbig = torch.zeros(n, h, w, 128, dtype=dt, device=dev)
    A.ops.conv_pair_raw(x, pk1, pk2, out=obig[..., 64:], mid=mbig[..., :64], relu_mid=True, scale_out=0.5, res=x)
    mid_ref, out_ref = _two_launch(A, x.contiguous(), pk1, pk2, relu_mid=True, scale_mid=1.0, mask=None, scale_out=0.5,
                                   res=x.contiguous(), use_bias=True)
    torch.cuda.synchronize()
    assert torch.equal(obig[..., 64:], out_ref) and torch.equal(mbig[..., :64], mid_ref)
    assert float(obig[..., :64].abs().max()) == 0.0 and float(mbig[..., 64:].abs().max()) == 0.0, "neighbouring channels untouched"


def test_pair_argument_errors(A):
    dev = torch.device("cuda")
    x = torch.zeros(1, 8, 8, 64, dtype=torch.float32, device=dev)
    w = torch.nn.Parameter(torch.zeros(64, 64, 3, 3, device=dev))
    pk = A.ops.pack_conv(w, None, torch.bfloat16)
    with pytest.raises(RuntimeError, match="16-bit"):
        A.ops.conv_pair_raw(x, pk, pk, out=torch.empty_like(x))
    xb = x.to(torch.bfloat16)
    with pytest.raises(RuntimeError, match="exclusive"):
        A.ops.conv_pair_raw(xb, pk, pk, out=torch.empty_like(xb), relu_mid=True, mask=xb)
    assert A._lib.load().srk_conv_pair_tiles(16, 48, 48) == 256
    assert A._lib.load().srk_conv_pair_tiles(1, 15, 14) == 2


def _block_grads(A, make, x0, paired):
    prev = A.ops._PAIR_OFF
    A.ops._PAIR_OFF = not paired
    try:
        torch.manual_seed(0)
        m = make()
        x = x0.clone().requires_grad_(True)
        y = m(x)
        (y.float() * torch.linspace(-1, 1, y.numel(), device=y.device).view_as(y)).sum().backward()
        torch.cuda.synchronize()
        return y.detach(), x.grad.detach(), [p.grad.detach().clone() for p in m.parameters()]
    finally:
        A.ops._PAIR_OFF = prev


@pytest.mark.parametrize("kind", ["resblock", "rcab"])
def test_blocks_take_the_pair_at_batch_16_and_agree_with_two_launches(A, kind, monkeypatch):
    """ResBlock / RCAB modules at the reference's 16 x 48 x 48: the Functions choose the pair; outputs and input gradient
    are bit-identical to the two-launch path, weight gradients agree to fp32 summation order."""
    dev = torch.device("cuda")
    dt = torch.bfloat16
    from sr_amd.models import common, rcan

    def make():
        if kind == "resblock":
            m = common.ResBlock(n_feats=64, kernel_size=3, res_scale=0.1)
        else:
            m = rcan.RCAB(common.DefaultConv2d, 64, 3, 16)
        m = m.to(dev)
        m.compute_dtype = dt          # fp32 parameters, bf16 activations (models/common._NCHWContract)
        return m

    calls = []
    real = A.ops.conv_pair_raw
    monkeypatch.setattr(A.ops, "conv_pair_raw", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    g = torch.Generator().manual_seed(2)
    x0 = _rand(g, 16, 64, 48, 48, dt=torch.float32, dev=dev)
    yp, gxp, gp = _block_grads(A, make, x0, True)
    assert len(calls) == 2, "one pair launch forward, one backward"
    yt, gxt, gt = _block_grads(A, make, x0, False)
    assert len(calls) == 2
    if kind == "resblock":
        assert torch.equal(yp, yt) and torch.equal(gxp, gxt)
    else:       # the pair also pools t for the channel attention: another summation order, a few outputs move by one bf16 ulp
        for a, b in ((yp, yt), (gxp, gxt)):
            d = (a - b).abs()
            assert float(d.max()) <= 2.0 ** -7 * float(b.abs().max()) and float((d > 0).float().mean()) < 0.02
    for a, b in zip(gp, gt):
        assert float((a.float() - b.float()).abs().max()) <= 1e-5 * float(b.float().abs().max()) + 1e-7


def test_pair_threshold_follows_tile_count(A):
    dev = torch.device("cuda")
    w = torch.zeros(64, 64, 3, 3, device=dev)
    cus = A._lib.load().srk_device_cus()
    small = torch.zeros(16, 48, 48, 64, dtype=torch.bfloat16, device=dev)
    assert A.ops.pair_ok(small, w, w)
    big = torch.zeros(256, 48, 48, 64, dtype=torch.bfloat16, device=dev)
    assert A._lib.load().srk_conv_pair_tiles(256, 48, 48) > 2 * cus and not A.ops.pair_ok(big, w, w)
    assert not A.ops.pair_ok(small.float(), w, w)
    assert not A.ops.pair_ok(small, torch.zeros(64, 64, 1, 1, device=dev), w)


def test_rcab_chain_pools_inside_the_pair_launches(A, monkeypatch):
    """Three RCABs in a row at 16 x 48 x 48: the forward pooling passes ride on the conv pairs, the backward ones (sum of
    t * gradient) on the dgrad pair of the FOLLOWING block; only the last block, whose gradient comes from elsewhere, still
    launches srk_ca_pool.  Results agree with the unfused path to bf16 rounding."""
    dev = torch.device("cuda")
    from sr_amd.models import common, rcan

    def make():
        m = torch.nn.Sequential(*[rcan.RCAB(common.DefaultConv2d, 64, 3, 16) for _ in range(3)]).to(dev)
        return m

    def run(paired):
        prev = A.ops._PAIR_OFF
        A.ops._PAIR_OFF = not paired
        try:
            torch.manual_seed(0)
            m = make()
            g = torch.Generator().manual_seed(4)
            x = A.ops.nchw_to_nhwc(_rand(g, 16, 64, 48, 48, dt=torch.float32, dev=dev), torch.bfloat16).detach().requires_grad_(True)
            y = x
            for blk in m:
                y = blk.nhwc(y)
            (y.float() * torch.linspace(-1, 1, y.numel(), device=dev).view_as(y)).sum().backward()
            torch.cuda.synchronize()
            return y.detach().float(), x.grad.float(), [p.grad.clone() for p in m.parameters()]
        finally:
            A.ops._PAIR_OFF = prev

    calls = []
    real = A._lib.call
    monkeypatch.setattr(A._lib, "call", lambda name, *a, **k: (calls.append(name), real(name, *a, **k))[1])
    yp, gxp, gp = run(True)
    assert calls.count("srk_conv_pair") == 6 and calls.count("srk_ca_pool") == 1, calls
    del calls[:]
    yt, gxt, gt = run(False)
    assert calls.count("srk_conv_pair") == 0 and calls.count("srk_ca_pool") == 6
    for a, b in ((yp, yt), (gxp, gxt)):
        d = (a - b).abs()
        assert float(d.max()) <= 2.0 ** -6 * float(b.abs().max()) and float((d > 0).float().mean()) < 0.05
    for a, b in zip(gp, gt):
        assert float((a - b).abs().max()) <= 2e-2 * float(b.abs().max()) + 1e-6


def test_rcab_hint_is_not_used_for_an_accumulated_gradient(A):
    """The output of an RCAB consumed twice: autograd sums two gradients, the sums pooled from one of them do not apply."""
    dev = torch.device("cuda")
    from sr_amd.models import common, rcan
    torch.manual_seed(1)
    b1 = rcan.RCAB(common.DefaultConv2d, 64, 3, 16).to(dev)
    b2 = rcan.RCAB(common.DefaultConv2d, 64, 3, 16).to(dev)
    g = torch.Generator().manual_seed(8)
    x0 = A.ops.nchw_to_nhwc(_rand(g, 4, 64, 20, 20, dt=torch.float32, dev=dev), torch.bfloat16).detach()

    def run(paired):
        prev = A.ops._PAIR_OFF
        A.ops._PAIR_OFF = not paired
        try:
            for p in list(b1.parameters()) + list(b2.parameters()):
                p.grad = None
            x = x0.clone().requires_grad_(True)
            y1 = b1.nhwc(x)
            y2 = b2.nhwc(y1)
            (y2.float().sum() + (y1.float() * 0.5).sum()).backward()        # y1 has two consumers
            torch.cuda.synchronize()
            return x.grad.float(), [p.grad.clone() for p in b1.parameters()]
        finally:
            A.ops._PAIR_OFF = prev

    gxp, gp = run(True)
    gxt, gt = run(False)
    assert float((gxp - gxt).abs().max()) <= 2.0 ** -6 * float(gxt.abs().max())
    for a, b in zip(gp, gt):
        assert float((a - b).abs().max()) <= 2e-2 * float(b.abs().max()) + 1e-6


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(16, 48, 48), (3, 20, 33), (2, 14, 14)])
def test_ca_backward_on_the_way_in_matches_its_own_launch(A, dt, shape):
    """ca_mode 1: srk_conv_pair applies the CALayer backward (models/rcan.py:10-29) to its input while loading it.  The
    transformed input gt, the per-sample parameter-gradient slots, the intermediate and the output must be BIT-identical to
    srk_ca_bwd_apply followed by the plain pair."""
    dev = torch.device("cuda")
    n, h, w = shape
    cr = 4
    g = torch.Generator().manual_seed(21)
    gin = _rand(g, n, h, w, 64, dt=dt, dev=dev)
    y1 = torch.relu(_rand(g, n, h, w, 64, dt=dt, dev=dev))
    rows_g, rows_s = 5, A._lib.load().srk_ca_splits(n, h * w)
    gsum = (torch.rand(n, rows_g, 64, generator=g) - 0.5).to(dev)
    sums = (torch.rand(n, rows_s, 64, generator=g) * (h * w / rows_s)).to(dev)
    sg = torch.sigmoid(torch.rand(n, 64, generator=g) * 4 - 2).to(dev)
    z = torch.relu(torch.rand(n, cr, generator=g) - 0.3).to(dev)
    w1f = ((torch.rand(cr, 64, generator=g) - 0.5) * 0.5).to(dev)
    w2f = ((torch.rand(64, cr, generator=g) - 0.5) * 0.5).to(dev)
    wa = torch.nn.Parameter((torch.rand(64, 64, 3, 3, generator=g) - 0.5).mul(0.1).to(dev))
    wb = torch.nn.Parameter((torch.rand(64, 64, 3, 3, generator=g) - 0.5).mul(0.1).to(dev))
    pka, pkb = A.ops.pack_conv(wa, None, dt, dgrad=True), A.ops.pack_conv(wb, None, dt, dgrad=True)
    K = 2 * 64 * cr + cr + 64
    L = A._lib
    # reference: its own launch, then the plain pair
    per_ref = torch.full((n, K), float("nan"), device=dev)
    gt_ref = torch.empty_like(gin)
    L.call("srk_ca_bwd_apply", L.CaBwdArgs(
        g=gin.data_ptr(), g_pitch=64, g_coff=0, gsum=gsum.data_ptr(), sums=sums.data_ptr(), s=sg.data_ptr(), z=z.data_ptr(),
        w1=w1f.data_ptr(), w2=w2f.data_ptr(), dw1=per_ref[0, :64 * cr].data_ptr(), db1=per_ref[0, 64 * cr:].data_ptr(),
        dw2=per_ref[0, 64 * cr + cr:].data_ptr(), db2=per_ref[0, 2 * 64 * cr + cr:].data_ptr(), gt=gt_ref.data_ptr(), gt_pitch=64, gt_coff=0,
        N=n, HW=h * w, C=64, Cr=cr, dtype=A.ops._DT[dt], sums_rows=rows_s, gsum_rows=rows_g), torch.cuda.current_stream().cuda_stream)
    mid_ref, out_ref = torch.empty_like(gin), torch.empty_like(gin)
    A.ops.conv_pair_raw(gt_ref, pka, pkb, out=out_ref, mask=y1, mid=mid_ref, res=gin, use_bias=False)
    # fused
    per = torch.full((n, K), float("nan"), device=dev)
    gt, mid, out = (torch.full_like(gin, float("nan")) for _ in range(3))
    A.ops.conv_pair_raw(gin, pka, pkb, out=out, mask=y1, mid=mid, res=gin, use_bias=False,
                        ca_bwd=dict(gsum=gsum, sums=sums, s=sg, z=z, w1=w1f, w2=w2f, slots=per), xo=gt)
    torch.cuda.synchronize()
    assert torch.equal(gt.view(torch.int16), gt_ref.view(torch.int16)), "gt"
    assert torch.equal(per, per_ref), "parameter-gradient slots"
    assert torch.equal(mid.view(torch.int16), mid_ref.view(torch.int16)), "intermediate"
    assert torch.equal(out.view(torch.int16), out_ref.view(torch.int16)), "output"


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(16, 48, 48), (3, 20, 33), (2, 14, 14)])
def test_ca_forward_on_the_way_in_matches_its_own_launch(A, dt, shape):
    """ca_mode 2: the previous block's `t * s + x` (srk_ca_apply) formed by the next block's conv launch while it loads its
    input: x', s, z, the intermediate and the output are BIT-identical to srk_ca_apply followed by the plain pair."""
    dev = torch.device("cuda")
    n, h, w = shape
    cr = 4
    g = torch.Generator().manual_seed(31)
    t = _rand(g, n, h, w, 64, dt=dt, dev=dev)
    xres = _rand(g, n, h, w, 64, dt=dt, dev=dev)
    rows = 7
    sums = ((torch.rand(n, rows, 64, generator=g) - 0.3) * (h * w / rows)).to(dev)
    w1f = ((torch.rand(cr, 64, generator=g) - 0.5) * 0.5).to(dev)
    w2f = ((torch.rand(64, cr, generator=g) - 0.5) * 2).to(dev)
    b1f = (torch.rand(cr, generator=g) - 0.5).to(dev)
    b2f = (torch.rand(64, generator=g) - 0.5).to(dev)
    wa = torch.nn.Parameter((torch.rand(64, 64, 3, 3, generator=g) - 0.5).mul(0.1).to(dev))
    wb = torch.nn.Parameter((torch.rand(64, 64, 3, 3, generator=g) - 0.5).mul(0.1).to(dev))
    ba = torch.nn.Parameter((torch.rand(64, generator=g) - 0.5).to(dev))
    bb = torch.nn.Parameter((torch.rand(64, generator=g) - 0.5).to(dev))
    pka, pkb = A.ops.pack_conv(wa, ba, dt), A.ops.pack_conv(wb, bb, dt)
    L = A._lib
    s_ref, z_ref = torch.empty(n, 64, device=dev), torch.empty(n, cr, device=dev)
    x_ref = torch.empty_like(t)
    L.call("srk_ca_apply", L.CaApplyArgs(
        t=t.data_ptr(), t_pitch=64, t_coff=0, res=xres.data_ptr(), res_pitch=64, res_coff=0, sums=sums.data_ptr(),
        w1=w1f.data_ptr(), b1=b1f.data_ptr(), w2=w2f.data_ptr(), b2=b2f.data_ptr(), s_out=s_ref.data_ptr(), z_out=z_ref.data_ptr(),
        out=x_ref.data_ptr(), out_pitch=64, out_coff=0, N=n, HW=h * w, C=64, Cr=cr, dtype=A.ops._DT[dt], sums_rows=rows),
        torch.cuda.current_stream().cuda_stream)
    mid_ref, out_ref = torch.empty_like(t), torch.empty_like(t)
    A.ops.conv_pair_raw(x_ref, pka, pkb, out=out_ref, relu_mid=True, mid=mid_ref)
    sv, zv = torch.full((n, 64), float("nan"), device=dev), torch.full((n, cr), float("nan"), device=dev)
    xo, mid, out = (torch.full_like(t, float("nan")) for _ in range(3))
    A.ops.conv_pair_raw(t, pka, pkb, out=out, relu_mid=True, mid=mid, xo=xo,
                        ca_fwd=dict(x2=xres, sums=sums, w1=w1f, b1=b1f, w2=w2f, b2=b2f, s_out=sv, z_out=zv))
    torch.cuda.synchronize()
    assert torch.equal(sv, s_ref) and torch.equal(zv, z_ref), "s / z"
    assert torch.equal(xo.view(torch.int16), x_ref.view(torch.int16)), "t * s + x"
    assert torch.equal(mid.view(torch.int16), mid_ref.view(torch.int16)), "intermediate"
    assert torch.equal(out.view(torch.int16), out_ref.view(torch.int16)), "output"


def test_residual_group_is_one_launch_per_block_and_direction(A, monkeypatch):
    """A residual group (models/rcan.py:59-74) of 4 RCABs at 16 x 48 x 48: forward 4 pair launches + 1 srk_ca_apply (the last
    block's), backward 4 pair launches + 1 srk_ca_pool (the last block's gradient comes from the group's conv); no
    srk_ca_bwd_apply at all.  Output and gradients agree with the launch-per-op path to bf16 rounding."""
    dev = torch.device("cuda")
    from sr_amd.models import common, rcan

    def run(fused):
        prev = A.ops._PAIR_OFF
        A.ops._PAIR_OFF = not fused
        try:
            torch.manual_seed(0)
            grp = rcan.ResidualGroup(common.DefaultConv2d, 64, 3, 16, act=None, res_scale=1, n_resblocks=4).to(dev)
            g = torch.Generator().manual_seed(4)
            x = A.ops.nchw_to_nhwc(_rand(g, 16, 64, 48, 48, dt=torch.float32, dev=dev), torch.bfloat16).detach().requires_grad_(True)
            y = grp.nhwc(x)
            (y.float() * torch.linspace(-1, 1, y.numel(), device=dev).view_as(y)).sum().backward()
            torch.cuda.synchronize()
            return y.detach().float(), x.grad.float(), [p.grad.clone() for p in grp.parameters()]
        finally:
            A.ops._PAIR_OFF = prev

    calls = []
    real = A._lib.call
    monkeypatch.setattr(A._lib, "call", lambda name, *a, **k: (calls.append(name), real(name, *a, **k))[1])
    yp, gxp, gp = run(True)
    cnt = {k: calls.count(k) for k in ("srk_conv_pair", "srk_ca_apply", "srk_ca_pool", "srk_ca_bwd_apply")}
    assert cnt == {"srk_conv_pair": 8, "srk_ca_apply": 1, "srk_ca_pool": 1, "srk_ca_bwd_apply": 0}, cnt
    del calls[:]
    yt, gxt, gt = run(False)
    assert calls.count("srk_conv_pair") == 0 and calls.count("srk_ca_apply") == 4 and calls.count("srk_ca_bwd_apply") == 4
    for a, b in ((yp, yt), (gxp, gxt)):
        d = (a - b).abs()
        assert float(d.max()) <= 2.0 ** -6 * float(b.abs().max()) and float((d > 0).float().mean()) < 0.05
    for a, b in zip(gp, gt):
        assert float((a - b).abs().max()) <= 2e-2 * float(b.abs().max()) + 1e-6
