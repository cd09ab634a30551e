"""Round-3 GPU parity tests: the batch-16 flavours of the path -- two convs per launch with the CALayer
steps inside the launches (`srk_conv_pair` ca_mode 1 / 2, models/rcan.py:10-74) -- compared with the ORACLE (the CPU restatement
pinned to the reference's own outputs), not with the stand-alone launches they replace.

All citations relative to /root/reference.
"""
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


@pytest.mark.parametrize("dt,min_psnr,min_cos", [(torch.bfloat16, 50.0, 0.99), (torch.float16, 62.0, 0.999)])
def test_rcan_64_feature_chain_16bit_vs_oracle(A, dt, min_psnr, min_cos):
    """BASELINE config 3's own shape per launch: RCAN with 64 features at 16 x 3 x 48 x 48 (one 14x14 tile per CU, the pair
    kernel with pooling, `ca_mode 2` forward and `ca_mode 1` backward through `ops.rcab_chain`), forward PSNR and the cosine
    of every sizeable parameter gradient against the fp32 oracle (rcan.py:33-74), plus the channel-attention parameters'
    gradients (conv_du: 64 -> 4 -> 64), which only the fused backward produces."""
    from sr_amd import ops
    torch.manual_seed(0)
    m = A.RCAN(precision=PREC[dt], **RCAN_KW)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    trainable = {k for k, p in m.named_parameters() if p.requires_grad}
    for k in trainable:
        sd[k].requires_grad_(True)
    m = m.cuda()
    gen = torch.Generator().manual_seed(777)
    x = torch.rand(16, 3, 48, 48, generator=gen)
    hr = torch.rand(16, 3, 96, 96, generator=gen)
    y_ref = OF.forward("RCAN", sd, x, **RCAN_KW)
    torch.nn.functional.l1_loss(y_ref, hr).backward()
    before = list(ops.PAIR_LAUNCHES)
    y = m(x.cuda())
    loss = torch.nn.functional.l1_loss(y, hr.cuda())
    (loss * 1024.0).backward()                      # loss scaling keeps fp16 gradients out of the subnormals
    torch.cuda.synchronize()
    took = [a - b for a, b in zip(ops.PAIR_LAUNCHES, before)]
    # 2 groups x 3 RCABs: forward = per group 1 plain + 2 with the previous block's CALayer forward (ca_mode 2);
    # backward = 6 launches with the CALayer backward on the way in (ca_mode 1)
    assert took[2] == 4 and took[1] == 6 and took[0] >= 2, f"pair launches by ca_mode: {took}"
    mse = float(((y.detach().cpu().double() - y_ref.detach().double()) ** 2).mean())
    psnr = 10 * np.log10(1.0 / max(mse, 1e-30))
    assert psnr > min_psnr, f"RCAN 64 {dt}: PSNR(build, oracle) = {psnr:.1f} dB"
    assert abs(float(loss) - float(torch.nn.functional.l1_loss(y_ref, hr))) < 2e-3
    params = dict(m.named_parameters())
    worst = 1.0
    for k in sorted(trainable):
        ref = sd[k].grad.double().flatten()
        if ref.numel() < 256:
            continue
        got = params[k].grad.cpu().double().flatten() / 1024.0
        cos = float(torch.dot(got, ref) / (got.norm() * ref.norm() + 1e-30))
        worst = min(worst, cos)
        assert cos > min_cos, f"RCAN 64 {dt} grad {k}: cosine {cos:.5f}"
        ratio = float(got.norm() / (ref.norm() + 1e-30))
        assert 0.9 < ratio < 1.1, f"RCAN 64 {dt} grad {k}: norm ratio {ratio:.3f}"
    ca_keys = [k for k in trainable if "conv_du" in k and k.endswith("weight")]
    assert len(ca_keys) == 12
    print(f"RCAN 64 {dt}: PSNR {psnr:.1f} dB, worst gradient cosine {worst:.5f}, pair launches {took}")


@pytest.mark.parametrize("cls,kw", [("RCAN", dict(n_feats=64, n_resgroups=2, n_resblocks=3, reduction=16, scale_factor=2)),
                                    ("EDSR", dict(n_feats=64, n_resblocks=4, res_scale=0.1, scale_factor=2)),
                                    # weight-normed convs: the effective weights are non-leaf tensors (ops.WeightNormGroup); a
                                    # reference to one that outlives its eager step used to crash the capture of a later step
                                    ("WDSR", dict(type="B", n_feats=64, n_resblocks=3, scale_factor=2)),
                                    ("WDSR", dict(type="A", n_feats=32, n_resblocks=2, scale_factor=2))])
def test_graph_replayed_steps_follow_the_oracle_trajectory(A, cls, kw):
    """Trainer.fit (3 eager steps, then hipGraph replays of the pair-kernel step) against the ORACLE's Adam trajectory on the
    same batches (srmodel.py:145-171): the loss of every step, computed from weights that all earlier steps produced."""
    from sr_amd import trainer as T
    torch.manual_seed(0)
    m = getattr(A, cls)(precision="bf16", **kw)
    om = OT.OracleModel(cls, **kw)
    om.load_state_dict({k: v.detach().clone() for k, v in m.state_dict().items()})
    steps = 8
    data = [T.synthetic_batch(16, 3, 48, 2, 500 + i, "cpu") for i in range(steps)]
    # learnable targets (HR = bilinear upsampling of LR) so that the loss moves by far more than the comparison tolerance
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
    assert tr.graphed is not None and tr.graphed.graphs is not None and not tr.graphed.failed
    got = tr.losses
    assert len(got) == steps
    assert ref[0] - ref[-1] > 0.05 * ref[0], f"the oracle's loss should fall on learnable data: {ref}"
    np.testing.assert_allclose(got, ref, rtol=2e-2, atol=2e-3)


def test_bench_runs_the_multi_rank_step_structure_on_one_gpu():
    """`SRK_FORCE_DDP=1 python bench.py --batch 16`: a 1-rank `nccl` (RCCL) group makes bench.py take the code every rank of
    `--gpus N` takes -- GradSync buckets with the gradients written straight into the flat buffer, forward + backward as one
    hipGraph, the bucket all-reduce and the optimizer step behind it -- and the line must say so and train (loss falls)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SRK_FORCE_DDP="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29731", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--batch", "16", "--steps", "30", "--warmup", "30", "--no-cpu-baseline",
                          "--no-roofline", "--no-other-configs", "--sustain-seconds", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["config"]["hip_graph"] == "segmented" and d["config"]["grad_sync"] == "bucketed_allreduce", d["config"]
    assert d["value"] > 1000 and d["config"]["loss_after_timed_steps"] < d["config"]["loss_after_warmup"]


def test_gradsync_gradients_land_in_the_flat_buffer(A):
    """trainer.GradSync names each parameter's slice of its flat gradient buffer as the weight-gradient kernels' target: after a
    backward pass the conv gradients ARE views of the buffer (nothing for pack() to copy) and equal the gradients of a run
    without GradSync."""
    from sr_amd import trainer as T
    torch.manual_seed(0)
    m = A.EDSR(n_feats=64, n_resblocks=2, res_scale=0.1, scale_factor=2, precision="bf16").cuda()
    b = T.synthetic_batch(4, 3, 24, 2, 11, "cuda")
    m.training_step(b, 0)["loss"].backward()
    ref = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    for p in m.parameters():
        p.grad = None
    gs = T.GradSync(m, overlap=False)
    m.training_step(b, 0)["loss"].backward()
    torch.cuda.synchronize()
    inplace = sum(1 for p in gs.params if p.grad is not None and p.grad.data_ptr() == gs.views[p].data_ptr())
    assert inplace >= len(gs.params) - 2, f"{inplace} of {len(gs.params)} gradients were written in place"
    gs.pack()
    for k, p in m.named_parameters():
        if k in ref:
            assert torch.equal(p.grad, ref[k]), k
    gs.detach()
    assert all("_srk_grad_target" not in p.__dict__ for p in m.parameters())


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
    orig = ops._unfold_raw
    ops._unfold_raw = lambda *a_, **k_: (calls.append(1), orig(*a_, **k_))[1]
    try:
        y = ops.conv_general(xd, wd, bd, stride=1, pad=k // 2)
        gyd = torch.zeros(n, h, w, y.shape[3], dtype=dt)
        gyd[..., :cout] = gy.permute(0, 2, 3, 1).to(dt)
        y.backward(gyd.cuda())
    finally:
        ops._unfold_raw = orig
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


def test_tiled_group_pack_is_bit_identical_to_the_standalone_pack(A):
    """ops.PackGroup.refresh (srk_pack_conv_weights_group_tiled: one 16 x 64-channel tile of one parameter per block, read through
    LDS) against srk_pack_conv_weights on the same parameters: every layout a model uses -- forward / dgrad, 3x3 / 1x1 / 5x5 (the
    latter through the 16-block fallback), channel counts that are not multiples of 8, 16 or 64, the PixelShuffle permutation on a
    forward layout (models/common.py Upsampler) and on a dgrad layout (fallback), bf16 and fp16 -- bit for bit, biases included,
    after the parameters changed in place."""
    from sr_amd import ops
    torch.manual_seed(11)
    shapes = [(64, 64, 3, 0), (64, 3, 3, 0), (3, 64, 3, 0), (256, 64, 3, 2), (102, 128, 3, 0), (128, 102, 3, 0), (48, 160, 1, 0),
              (64, 320, 1, 0), (12, 3, 5, 0), (36, 64, 3, 3), (200, 72, 3, 0)]
    for dt in (torch.bfloat16, torch.float16):
        params = [(torch.nn.Parameter(torch.randn(co, ci, k, k, device="cuda") * 0.1), torch.nn.Parameter(torch.randn(co, device="cuda")), ps)
                  for (co, ci, k, ps) in shapes]
        grp = ops.PackGroup()
        with ops.forward_scope(grp):
            held = [(ops.pack_conv(w, b, dt, ps_r=ps), ops.pack_conv(w, None, dt, dgrad=True, ps_r=ps)) for (w, b, ps) in params]
        with torch.no_grad():
            for (w, b, _) in params:                              # an optimizer step: new values at the same addresses
                w.mul_(-0.7).add_(0.01)
                b.add_(0.5)
        with ops.forward_scope(grp):                              # the grouped launch refreshes every layout in place
            again = [(ops.pack_conv(w, b, dt, ps_r=ps), ops.pack_conv(w, None, dt, dgrad=True, ps_r=ps)) for (w, b, ps) in params]
        torch.cuda.synchronize()
        for (w, b, ps), (pf, pd), (qf, qd), shp in zip(params, held, again, shapes):
            assert qf.wpk.data_ptr() == pf.wpk.data_ptr() and qd.wpk.data_ptr() == pd.wpk.data_ptr(), "the group serves its own buffers"
            rf = ops.pack_conv(w.detach().clone().requires_grad_(False), b.detach().clone(), dt, ps_r=ps, cache=False)
            rd = ops.pack_conv(w.detach().clone(), None, dt, dgrad=True, ps_r=ps, cache=False)
            torch.cuda.synchronize()
            assert torch.equal(qf.wpk.view(torch.int16), rf.wpk.view(torch.int16)), ("forward", shp, dt)
            assert torch.equal(qf.bias, rf.bias), ("bias", shp, dt)
            assert torch.equal(qd.wpk.view(torch.int16), rd.wpk.view(torch.int16)), ("dgrad", shp, dt)


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
