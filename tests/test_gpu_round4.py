"""GPU tests added in round 4: the fp16 training step under hipGraph with device-resident dynamic loss scaling (BASELINE config 5's
dtype; the reference's `precision: 16` = Lightning "16-mixed" = autocast + torch.amp.GradScaler, configs/all.yml:122), re-capture after
a hyper-parameter change, WDSR under the multi-rank graph step, shared weights under GradSync."""
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


def test_device_grad_scaler_follows_torch_gradscaler_and_skips_on_inf(A):
    """optim.Adam.step(grad_scaler=DeviceGradScaler) against torch.optim.Adam under torch.amp.GradScaler on the same scaled gradients:
    same parameters after every step, a step with an injected inf changes NOTHING (parameters, moments, step counts) and halves the
    scale, the scale grows after `growth_interval` clean steps -- and no host read happens inside step()."""
    ps, rs = _params(3), _params(3)
    opt, ropt = A.optim.Adam(ps, lr=1e-2), torch.optim.Adam(rs, lr=1e-2)
    sc = A.optim.DeviceGradScaler("cuda", init_scale=1024.0, growth_interval=3)
    rsc = torch.amp.GradScaler("cuda", init_scale=1024.0, growth_interval=3)
    rsc.scale(torch.zeros(1, device="cuda"))          # (GradScaler creates its device state on the first scale())
    bad_step = 2
    for step in range(8):
        s_now = sc.get_scale()
        assert s_now == rsc.get_scale(), (step, s_now, rsc.get_scale())
        _grads(ps, step, s_now)
        _grads(rs, step, s_now)
        if step == bad_step:
            ps[1].grad.view(-1)[123] = float("inf")
            rs[1].grad.view(-1)[123] = float("inf")
        before = [p.detach().clone() for p in ps]
        opt.step(grad_scaler=sc)
        # torch's flow: unscale_ + inf check + (maybe) step + update
        rsc.unscale_(ropt)
        rsc.step(ropt)
        rsc.update()
        torch.cuda.synchronize()
        if step == bad_step:
            for p, b in zip(ps, before):
                assert torch.equal(p.detach(), b)
            assert sc.skipped_steps == 1
        for p, r in zip(ps, rs):
            assert float((p.detach() - r.detach()).abs().max()) <= 2e-6 * max(1.0, float(r.detach().abs().max())), (step, p.shape)
    assert float(opt.state[ps[0]]["step"]) == 7.0          # the skipped step did not count
    assert sc.get_scale() == rsc.get_scale()


@pytest.mark.parametrize("cls,kw", [("EDSR", dict(n_feats=64, n_resblocks=3, res_scale=0.1, scale_factor=2)),
                                    ("RCAN", dict(n_feats=64, n_resgroups=1, n_resblocks=3, reduction=16, scale_factor=2))])
def test_fp16_trainer_replays_a_graph_and_follows_the_oracle(A, cls, kw):
    """Trainer.fit of an fp16 model: the step IS captured (round 3: the GradScaler switched the graph off and config 5's dtype
    trained launch by launch) and the losses follow the fp32 oracle's Adam trajectory on the same batches."""
    from sr_amd import trainer as T
    torch.manual_seed(0)
    m = getattr(A, cls)(precision=16, **kw)
    om = OT.OracleModel(cls, **kw)
    om.load_state_dict({k: v.detach().clone() for k, v in m.state_dict().items()})
    steps = 8
    data = [T.synthetic_batch(16, 3, 48, 2, 900 + i, "cpu") for i in range(steps)]
    for b in data:
        b["hr"] = torch.nn.functional.interpolate(b["lr"], scale_factor=2, mode="bilinear", align_corners=False)
    opt = om.configure_optimizers()[0]
    ref = []
    for b in data:
        opt.zero_grad()
        loss = om.training_step(b)["loss"]
        loss.backward()
        opt.step()
        ref.append(float(loss))
    tr = T.Trainer(device="cuda", use_graph=True)
    tr.fit(m, iter(data))
    torch.cuda.synchronize()
    assert tr.scaler is not None and hasattr(tr.scaler, "state"), "fp16 must train under the device-resident loss scale"
    assert tr.graphed is not None and tr.graphed.graphs is not None and not tr.graphed.failed
    got = tr.losses
    assert len(got) == steps and all(np.isfinite(got))
    assert tr.scaler.skipped_steps == 0
    np.testing.assert_allclose(got, ref, rtol=2e-2, atol=2e-3)


def test_fp16_graph_replay_skips_a_step_with_an_injected_inf(A):
    """A replayed fp16 step whose input makes the gradients overflow: the replay itself (no Python in between) leaves the weights
    untouched and halves the scale; the next clean replay trains again."""
    from sr_amd import trainer as T
    torch.manual_seed(0)
    m = A.EDSR(n_feats=64, n_resblocks=2, res_scale=0.1, scale_factor=2, precision=16).cuda()
    opt = m.configure_optimizers()[0]
    sc = A.optim.DeviceGradScaler("cuda")
    gs = T.GraphedStep(m, m, opt, None, warm_steps=2, scaler=sc)
    good = T.synthetic_batch(4, 3, 24, 2, 5, "cuda")
    for _ in range(4):
        gs(good)
    torch.cuda.synchronize()
    assert gs.graphs is not None and not gs.failed
    bad = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in good.items()}
    bad["hr"][0, 0, 0, 0] = float("inf")             # |sr - hr| = inf: the L1 sign map is finite, so poison the input image instead
    bad["lr"][0, 0, 0, 0] = 6.0e4                    # fp16 activations overflow downstream of this pixel
    w0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    s0, k0 = sc.get_scale(), sc.skipped_steps
    gs(bad)
    torch.cuda.synchronize()
    if sc.skipped_steps == k0 + 1:                   # the overflow reached a gradient: nothing may have moved
        for k, v in m.state_dict().items():
            assert torch.equal(v, w0[k]), k
        assert sc.get_scale() == s0 * 0.5
    else:                                            # (no overflow on this input: the step was an ordinary one)
        assert all(torch.isfinite(v).all() for v in m.state_dict().values())
    gs(good)
    torch.cuda.synchronize()
    assert all(torch.isfinite(v).all() for v in m.state_dict().values())


def test_graph_recapture_after_lr_change_in_single_process_form(A):
    """changing the learning rate after the first replay re-captures the step; the capture must find a table reserved
    OUTSIDE the capture (page-locked memory cannot be allocated inside one) -- it used to fail and training stayed eager."""
    from sr_amd import trainer as T
    torch.manual_seed(0)
    m = A.EDSR(n_feats=64, n_resblocks=2, res_scale=0.1, scale_factor=2, precision="bf16").cuda()
    opt = m.configure_optimizers()[0]
    gs = T.GraphedStep(m, m, opt, None, warm_steps=2)
    b = T.synthetic_batch(4, 3, 24, 2, 5, "cuda")
    for _ in range(5):
        gs(b)
    assert gs.graphs is not None and not gs.failed
    g_before = gs.graphs[0]
    opt.param_groups[0]["lr"] = 5e-4
    for _ in range(3):
        gs(b)
    torch.cuda.synchronize()
    assert gs.graphs is not None and not gs.failed and gs.graphs[0] is not g_before
    assert all(len(p.captured) <= 1 for p in opt._plans.values()), "tables of replaced graphs must be released"
    # and the new rate is the one in effect: one more step moves the weights by about lr, not 1e-3
    w0 = m.head[0].weight.detach().clone()
    gs(b)
    torch.cuda.synchronize()
    d = float((m.head[0].weight.detach() - w0).abs().max())
    assert 0 < d <= 5.5e-4, d


def test_shared_conv_under_gradsync_gets_the_sum_of_both_uses(A):
    """a conv used TWICE in one forward while trainer.GradSync names its flat-buffer slice as the gradient target: the second
    use must not overwrite the first in the same memory (autograd would then add two aliases: 2 g_2 instead of g_1 + g_2)."""
    from sr_amd import trainer as T, ops

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c = torch.nn.Conv2d(64, 64, 5, padding=2)       # 5x5: the immediate (not deferred) weight-gradient path

        def forward(self, x):
            y = ops.conv_general(x, self.c.weight, self.c.bias, stride=1, pad=2)
            return ops.conv_general(y, self.c.weight, self.c.bias, stride=1, pad=2)

    torch.manual_seed(0)
    net = Net().cuda()
    x = (torch.rand(2, 64, 20, 20, device="cuda") - 0.5)

    def run():
        for p in net.parameters():
            p.grad = None
        xh = x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)
        net(xh).float().square().mean().backward()
        torch.cuda.synchronize()
        return {k: p.grad.detach().clone() for k, p in net.named_parameters()}
    ref = run()
    gs = T.GradSync(net, overlap=False)
    got = run()
    gs.detach()
    for k in ref:
        assert torch.allclose(got[k], ref[k], rtol=1e-5, atol=1e-7), k


def test_wdsr_keeps_one_backward_graph_under_the_multi_rank_step(A):
    """WDSR's 51 weight norms are one autograd node; the segmented backward would run it twice.  auto_segments answers 1 for
    it whatever SRK_DDP_SEGMENTS says, so GraphedStep keeps the single backward graph."""
    from sr_amd import trainer as T
    m = A.WDSR(type="B", n_feats=64, n_resblocks=2, scale_factor=2, precision="bf16")
    os.environ["SRK_DDP_SEGMENTS"] = "3"
    try:
        assert T.auto_segments(m) == 1
        assert T.auto_segments(A.EDSR(n_feats=64, n_resblocks=2, scale_factor=2)) == 3
    finally:
        os.environ.pop("SRK_DDP_SEGMENTS", None)


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
@pytest.mark.parametrize("shape", [(40, 48, 48), (3, 17, 29)])
def test_relu_sign_bits_equal_the_activation_mask(A, dt, shape):
    """The ReLU backward of ResBlock / RCAB (models/common.py:99-100, rcan.py:33-55) masks with 4 bytes of sign bits per pixel and
    32-channel half, written by the forward conv + ReLU launch (srk_conv_args.relu_bits / mask_bits), instead of re-reading the 64-byte
    activation: every gradient must be BIT-identical to the activation-mask form, and the bits must be the activation's signs."""
    from sr_amd import ops
    n, h, w = shape
    torch.manual_seed(1)
    x = ((torch.rand(n, h, w, 64, device="cuda") - 0.5) * 2).to(dt)
    ws = [torch.nn.Parameter((torch.rand(64, 64, 3, 3, device="cuda") - 0.5) * 0.08) for _ in range(2)]
    bs = [torch.nn.Parameter((torch.rand(64, device="cuda") - 0.5) * 0.1) for _ in range(2)]
    g = ((torch.rand(n, h, w, 64, device="cuda") - 0.5)).to(dt)
    if ops.pair_ok(x, ws[0], ws[1]):
        pytest.skip("this batch takes the pair kernel")

    def run():
        xx = x.clone().requires_grad_(True)
        for p in ws + bs:
            p.grad = None
        y = ops.conv_chain(xx, [(ws[0], bs[0]), (ws[1], bs[1])], [True, False], scale=0.1)
        y.backward(g)
        torch.cuda.synchronize()
        return y.detach().clone(), xx.grad.clone(), [p.grad.clone() for p in ws + bs]
    assert ops._SIGN_BITS
    # the producer's bits against the stored activation
    pk = ops.pack_conv(ws[0], bs[0], dt)
    mid = torch.empty_like(x)
    ops.conv_raw(x, pk, N=n, H=h, W=w, Cin=64, Cout=64, out=mid, relu=True, relu_bits="want")
    bits = mid.__dict__.pop("_srk_bits")
    assert bits is not None and tuple(bits.shape) == (n * h * w, 2)
    torch.cuda.synchronize()
    m = (mid.float().view(-1, 2, 16, 2) > 0)                      # [pixel][half][dword i][lo / hi]
    want = (m[..., 0].long() << torch.arange(16, device="cuda")).sum(-1) + (m[..., 1].long() << (torch.arange(16, device="cuda") + 16)).sum(-1)
    assert torch.equal(bits.long() & 0xffffffff, want)
    y1, gx1, gp1 = run()
    ops._SIGN_BITS = False
    try:
        y2, gx2, gp2 = run()
    finally:
        ops._SIGN_BITS = True
    assert torch.equal(y1, y2) and torch.equal(gx1, gx2)
    for u, v in zip(gp1, gp2):
        assert torch.equal(u, v)


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
