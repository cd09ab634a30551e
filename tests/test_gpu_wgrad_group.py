"""GPU tests of the deferred, grouped weight gradients (ops.wgrad / ops.flush_wgrads -> srk_conv2d_wgrad_group +
srk_wgrad_finalize_group): one dispatch for all 3x3 weight gradients of a backward pass must give what the per-layer
launches give (same slab kernel body, different slab partition: sums differ by fp32 rounding only), match a float64
reference, be bitwise reproducible, and keep autograd's semantics (accumulation into existing .grad, a weight used twice,
non-leaf weights, hooks)."""
import ctypes as C
import os
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def A():
    import sr_amd
    assert torch.cuda.is_available()
    sr_amd._lib.load()
    return sr_amd


def _ref_wgrad(x, dy, k=3):
    """float64 weight/bias gradient of a 'same' conv from NHWC 16-bit operands."""
    xd = x.double().cpu().permute(0, 3, 1, 2).requires_grad_(False)
    dyd = dy.double().cpu().permute(0, 3, 1, 2)
    w = torch.zeros(dyd.shape[1], xd.shape[1], k, k, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(dyd.shape[1], dtype=torch.float64, requires_grad=True)
    y = F.conv2d(xd, w, b, padding=k // 2)
    (y * dyd).sum().backward()
    return w.grad, b.grad


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_grouped_matches_per_layer_and_float64(A, dt):
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(3)
    shapes = [(4, 48, 48, 64, 64), (2, 40, 23, 64, 64), (1, 96, 96, 64, 16), (3, 17, 50, 64, 256), (2, 33, 31, 128, 64)]
    jobs = []
    for (n, h, w, ci, co) in shapes:
        x = (torch.rand(n, h, w, ci, generator=g) - 0.5).to(dt).to(dev)
        dy = (torch.rand(n, h, w, co, generator=g) - 0.5).to(dt).to(dev)
        wp = torch.nn.Parameter(torch.zeros(co, ci, 3, 3, device=dev))
        bp = torch.nn.Parameter(torch.zeros(co, device=dev))
        kw = dict(N=n, H=h, W=w, Cin=ci, Cout=co, k=3, w_shape=(co, ci, 3, 3), scale=0.5 if co == 16 else 1.0)
        jobs.append((x, dy, wp, bp, kw))
    with A.ops.hold_wgrads():
        got = [A.ops.wgrad(x, dy, wparam=wp, bparam=bp, **kw) for (x, dy, wp, bp, kw) in jobs]
        assert len(A.ops._WQ.jobs) == len(jobs), "all five are 3x3 16-bit leaf-parameter jobs: queued, not launched"
    assert not A.ops._WQ.jobs
    torch.cuda.synchronize()
    for (x, dy, wp, bp, kw), (gw, gb) in zip(jobs, got):
        rw, rb = A.ops.wgrad_raw(x, dy, **kw)
        sc = kw["scale"]
        assert float((gw - rw).abs().max()) <= 2e-5 * float(rw.abs().max())
        assert float((gb - rb).abs().max()) <= 2e-5 * float(rb.abs().max()) + 1e-6
        fw, fb = _ref_wgrad(x, dy)
        assert float((gw.double().cpu() - sc * fw).abs().max()) <= 1e-4 * float(fw.abs().max())
        assert float((gb.double().cpu() - sc * fb).abs().max()) <= 1e-4 * float(fb.abs().max())


def _small_edsr(A, prec="bf16"):
    torch.manual_seed(0)
    return A.EDSR(n_feats=64, n_resblocks=3, res_scale=0.1, scale_factor=2, precision=prec).cuda()


def _grads(m, lr, hr):
    for p in m.parameters():
        p.grad = None
    m._calculate_losses(img_sr=m(lr), img_hr=hr)["loss"].backward()
    torch.cuda.synchronize()
    return {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}


def test_model_backward_deferred_equals_immediate_and_is_reproducible(A):
    m = _small_edsr(A)
    g = torch.Generator().manual_seed(1)
    lr, hr = torch.rand(3, 3, 40, 36, generator=g).cuda(), torch.rand(3, 3, 80, 72, generator=g).cuda()
    d1 = _grads(m, lr, hr)
    d2 = _grads(m, lr, hr)
    prev = A.ops.set_defer_wgrad(False)
    try:
        im = _grads(m, lr, hr)
    finally:
        A.ops.set_defer_wgrad(prev)
    assert set(d1) == set(im) and len(d1) > 10
    for k in d1:
        assert torch.equal(d1[k], d2[k]), f"{k}: the grouped launch is not bitwise reproducible"
        assert float((d1[k] - im[k]).abs().max()) <= 5e-5 * float(im[k].abs().max()) + 1e-9, k


def test_accumulates_into_existing_grads(A):
    m = _small_edsr(A)
    g = torch.Generator().manual_seed(2)
    lr, hr = torch.rand(2, 3, 24, 24, generator=g).cuda(), torch.rand(2, 3, 48, 48, generator=g).cuda()
    one = _grads(m, lr, hr)
    m._calculate_losses(img_sr=m(lr), img_hr=hr)["loss"].backward()      # second backward, .grad kept: += in the finalize
    torch.cuda.synchronize()
    for k, p in m.named_parameters():
        if p.grad is not None:
            assert float((p.grad - 2 * one[k]).abs().max()) <= 1e-5 * float(one[k].abs().max()) + 1e-9, k


def test_weight_used_twice_in_one_pass(A):
    dev = torch.device("cuda")
    torch.manual_seed(5)
    w = torch.nn.Parameter(torch.randn(64, 64, 3, 3, device=dev) * 0.03)
    b = torch.nn.Parameter(torch.randn(64, device=dev) * 0.1)
    x = (torch.rand(2, 20, 20, 64, device=dev) - 0.5).to(torch.bfloat16)

    def run():
        w.grad = b.grad = None
        y = A.ops.conv(A.ops.conv(x, w, b), w, b)
        y.float().square().mean().backward()
        torch.cuda.synchronize()
        return w.grad.clone(), b.grad.clone()
    gw, gb = run()
    prev = A.ops.set_defer_wgrad(False)
    try:
        rw, rb = run()
    finally:
        A.ops.set_defer_wgrad(prev)
    assert float((gw - rw).abs().max()) <= 5e-5 * float(rw.abs().max())
    assert float((gb - rb).abs().max()) <= 5e-5 * float(rb.abs().max())


def test_non_leaf_weight_and_hooked_parameter_are_not_deferred(A):
    dev = torch.device("cuda")
    torch.manual_seed(6)
    v = torch.nn.Parameter(torch.randn(64, 64, 3, 3, device=dev) * 0.03)
    x = (torch.rand(1, 16, 16, 64, device=dev) - 0.5).to(torch.bfloat16)
    seen = []
    y = A.ops.conv(x, v * 2.0, None)                 # weight is a non-leaf (weight-norm style): its gradient is consumed in-pass
    y.float().sum().backward()
    torch.cuda.synchronize()
    ref = v.grad.clone()
    v.grad = None
    h = v.register_hook(lambda g_: seen.append(float(g_.abs().sum())))   # a tensor hook reads the gradient inside backward
    A.ops.conv(x, v, None).float().sum().backward()
    torch.cuda.synchronize()
    h.remove()
    assert seen and abs(seen[0] - float(v.grad.abs().sum())) <= 1e-3 * seen[0]
    assert float((2.0 * v.grad - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


def test_upload_small_roundtrip(A):
    lib = A._lib.load()
    for n in (1, 15, 16, 3583, 3584, 3585, 20000):
        src = torch.randint(0, 256, (n,), dtype=torch.uint8)
        dst = torch.zeros(n + 32, dtype=torch.uint8, device="cuda")
        buf = (C.c_ubyte * n).from_buffer_copy(src.numpy().tobytes())
        rc = lib.srk_upload_small(dst.data_ptr(), C.addressof(buf), n, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(dst[:n].cpu(), src) and int(dst[n:].sum()) == 0


def test_channel_attention_gradients_are_summed_by_one_deferred_launch(A, monkeypatch):
    """RCAB backward: the per-sample slots of the conv_du gradients of every block are summed over the batch by ONE
    srk_rowsum_group launch when the pass ends; a second pass accumulates into the existing .grad through autograd
    (immediate sums), and a module applied twice in one pass is handled."""
    dev = torch.device("cuda")
    from sr_amd.models import common, rcan
    torch.manual_seed(3)
    blocks = [rcan.RCAB(common.DefaultConv2d, 64, 3, 16).to(dev) for _ in range(3)]
    g = torch.Generator().manual_seed(12)
    x0 = (torch.rand(8, 40, 40, 64, generator=g) - 0.5).to(torch.bfloat16).to(dev)
    names = []
    real = A._lib.load().srk_rowsum_group

    def run(order):
        x = x0.clone().requires_grad_(True)
        y = x
        for i in order:
            y = blocks[i].nhwc(y)
        y.float().square().sum().backward()
        torch.cuda.synchronize()

    def ca_grads():
        return [p.grad.clone() for b in blocks for p in b.body[3].parameters()]

    calls = []
    lib = A._lib.load()
    monkeypatch.setattr(A.ops, "_launch_rowsums", (lambda f: (lambda rj, st: (calls.append(len(rj)), f(rj, st))[1]))(A.ops._launch_rowsums))
    run([0, 1, 2])
    assert calls == [3], "one launch for the three blocks"
    g1 = ca_grads()
    prev = A.ops.set_defer_wgrad(False)
    try:
        for b in blocks:
            b.zero_grad(set_to_none=True)
        run([0, 1, 2])
    finally:
        A.ops.set_defer_wgrad(prev)
    g0 = ca_grads()
    for a, b in zip(g1, g0):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-7
    # second pass on top of existing gradients: accumulated, twice the value
    del calls[:]
    for b in blocks:
        for p in b.body[3].parameters():
            p.grad = None
    run([0, 1, 2])
    run([0, 1, 2])
    assert calls == [3], "the second pass finds gradients in place and sums immediately"
    for a, b in zip(ca_grads(), g1):
        assert float((a - 2 * b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-7
    # a block applied twice in one pass
    for b in blocks:
        b.zero_grad(set_to_none=True)
    run([0, 0])
    ga = [p.grad.clone() for p in blocks[0].body[3].parameters()]
    prev = A.ops.set_defer_wgrad(False)
    try:
        for b in blocks:
            b.zero_grad(set_to_none=True)
        run([0, 0])
    finally:
        A.ops.set_defer_wgrad(prev)
    for a, b in zip(ga, [p.grad for p in blocks[0].body[3].parameters()]):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-7


TH_CHILD = r'''
import json, sys, torch
sys.path.insert(0, %r)
import sr_amd as A
ops = A.ops
out = {}
for name, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
    for (n, h, w, cin, cout) in ((3, 33, 20, 64, 64), (5, 7, 9, 64, 64), (16, 48, 48, 64, 64), (2, 24, 24, 64, 256), (2, 17, 16, 128, 64), (1, 8, 8, 64, 64)):
        g = torch.Generator().manual_seed(n * 1000 + h)
        x = (torch.rand(n, h, w, cin, generator=g) - 0.5).to(dt).cuda()
        dy = (torch.rand(n, h, w, cout, generator=g) - 0.5).to(dt).cuda()
        ws = [torch.nn.Parameter(torch.zeros(cout, cin, 3, 3, device="cuda")) for _ in range(3)]
        bs = [torch.nn.Parameter(torch.zeros(cout, device="cuda")) for _ in range(3)]
        res = []
        with ops.hold_wgrads():                                   # the grouped launch (three jobs)
            for wi, bi in zip(ws, bs):
                res.append(ops.wgrad(x, dy, wparam=wi, bparam=bi, N=n, H=h, W=w, Cin=cin, Cout=cout, k=3, w_shape=(cout, cin, 3, 3)))
        gw1, gb1 = ops.wgrad_raw(x, dy, N=n, H=h, W=w, Cin=cin, Cout=cout, k=3, w_shape=(cout, cin, 3, 3))      # the single launch
        torch.cuda.synchronize()
        ref = torch.nn.grad.conv2d_weight(x.double().cpu().permute(0, 3, 1, 2), (cout, cin, 3, 3), dy.double().cpu().permute(0, 3, 1, 2), padding=1)
        bref = dy.double().cpu().sum((0, 1, 2))
        gw, gb = res[0][0].detach().double().cpu(), res[0][1].detach().double().cpu()
        out[f"{name}_{n}x{h}x{w}_{cin}_{cout}"] = {
            "rel": float((gw - ref).norm() / ref.norm()), "brel": float((gb - bref).norm() / bref.norm()),
            "rel_single": float((gw1.double().cpu() - ref).norm() / ref.norm()), "same_jobs": bool(torch.equal(res[0][0], res[2][0])),
            "sum": float(gw.sum()), "abs": float(gw.abs().sum())}
print("RESULT " + json.dumps(out))
''' % ROOT


def _th_run(env_extra):
    import json as _json
    import subprocess as _sp
    env = dict(os.environ, **env_extra)
    p = _sp.run([sys.executable, "-c", TH_CHILD], capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    return _json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])


def test_both_tile_geometries_of_the_slab_weight_gradient_against_float64():
    """Round 6: the slab-mode 3x3 weight gradient cuts the images into 8-row tiles (a ring of four buffers, three tiles in flight) where a workgroup walks many
    tiles, and into 16-row tiles (two buffers) otherwise (csrc/conv_wgrad.hip: ws_tile_height).  Both forms, forced through SRK_WGRAD_TH, on ragged images, one-tile
    images, 128 input / 256 output channels, grouped and single launch: each within fp32 rounding of the float64 gradient of the rounded operands, and of each other."""
    r8 = _th_run({"SRK_DEBUG": "1", "SRK_WGRAD_TH": "8"})
    r16 = _th_run({"SRK_DEBUG": "1", "SRK_WGRAD_TH": "16"})
    assert set(r8) == set(r16) and len(r8) == 12
    for k in r8:
        for r in (r8[k], r16[k]):
            assert r["rel"] < 5e-5 and r["rel_single"] < 5e-5 and r["brel"] < 5e-5 and r["same_jobs"], (k, r)
        assert abs(r8[k]["sum"] - r16[k]["sum"]) <= 1e-5 * r16[k]["abs"], k


def test_large_batches_take_the_ring_automatically(A):
    """1,024 16-row tiles per convolution switch the launch to the 8-row ring without any knob: a batch of 1,024 16 x 16 images against float64."""
    n, h, w, c = 1024, 16, 16, 64
    g = torch.Generator().manual_seed(5)
    x = (torch.rand(n, h, w, c, generator=g) - 0.5).bfloat16().cuda()
    dy = (torch.rand(n, h, w, c, generator=g) - 0.5).bfloat16().cuda()
    gw, gb = A.ops.wgrad_raw(x, dy, N=n, H=h, W=w, Cin=c, Cout=c, k=3, w_shape=(c, c, 3, 3))
    torch.cuda.synchronize()
    ref = torch.nn.grad.conv2d_weight(x.double().cpu().permute(0, 3, 1, 2), (c, c, 3, 3), dy.double().cpu().permute(0, 3, 1, 2), padding=1)
    assert float((gw.double().cpu() - ref).norm() / ref.norm()) < 5e-5
    assert float((gb.double().cpu() - dy.double().cpu().sum((0, 1, 2))).norm() / dy.double().cpu().sum((0, 1, 2)).norm()) < 5e-5
