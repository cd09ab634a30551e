"""GPU tests added in round 5 (ADVICE r4 / VERDICT r4): the device-resident loss scaler with several parameter groups, the
gradient-target alias guard across backward segments, conv_bits_ok's refusals, conv_pair's split epilogue flavours."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    import sr_amd
    assert torch.cuda.is_available()
    sr_amd._lib.load()
    return sr_amd


def _params(seed, shapes):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter((torch.rand(*s, generator=g) - 0.5).cuda()) for s in shapes]


def test_device_grad_scaler_checks_every_group_before_any_update(A):
    """with TWO parameter groups an inf in group 1 must leave group 0 untouched too (torch.amp.GradScaler unscales and
    checks every group, then steps or skips the whole optimizer); the per-group launcher used to update group 0 first."""
    shapes0, shapes1 = [(5,), (33, 7)], [(4097,), (8, 3, 3, 3)]
    p0, p1 = _params(1, shapes0), _params(2, shapes1)
    r0, r1 = _params(1, shapes0), _params(2, shapes1)
    opt = A.optim.Adam([{"params": p0, "lr": 1e-2}, {"params": p1, "lr": 3e-3}])
    ropt = torch.optim.Adam([{"params": r0, "lr": 1e-2}, {"params": r1, "lr": 3e-3}])
    sc = A.optim.DeviceGradScaler("cuda", init_scale=256.0, growth_interval=100)
    rsc = torch.amp.GradScaler("cuda", init_scale=256.0, growth_interval=100)
    rsc.scale(torch.zeros(1, device="cuda"))
    for step in range(5):
        s_now = sc.get_scale()
        assert s_now == rsc.get_scale()
        g = torch.Generator().manual_seed(50 + step)
        for p, r in zip(p0 + p1, r0 + r1):
            gr = ((torch.rand(*p.shape, generator=g) - 0.5) * s_now).cuda()
            p.grad, r.grad = gr.clone(), gr.clone()
        if step == 2:                                        # the LAST group carries the inf
            p1[0].grad[77] = float("inf")
            r1[0].grad[77] = float("inf")
        before = [p.detach().clone() for p in p0 + p1]
        steps_before = [float(opt.state[p]["step"]) if "step" in opt.state[p] else 0.0 for p in p0 + p1]
        opt.step(grad_scaler=sc)
        rsc.unscale_(ropt)
        rsc.step(ropt)
        rsc.update()
        torch.cuda.synchronize()
        if step == 2:
            for p, b in zip(p0 + p1, before):
                assert torch.equal(p.detach(), b), "a skipped step changed a parameter"
            assert [float(opt.state[p]["step"]) for p in p0 + p1] == steps_before
            assert sc.skipped_steps == 1
        for p, r in zip(p0 + p1, r0 + r1):
            assert float((p.detach() - r.detach()).abs().max()) <= 2e-6 * max(1.0, float(r.detach().abs().max())), (step, p.shape)
    assert sc.get_scale() == rsc.get_scale()


def test_conv_bits_ok_refuses_pixel_shuffle_and_post_add(A):
    """the sign-bit store indexes its [N*H*W][2] buffer by the OUTPUT pixel: a pixel-shuffled store (ps_r > 1) or a planar
    epilogue with post_add is not a bits launch -- srk_conv_bits_ok is the gate the C ABI advertises and must say so."""
    L = A._lib
    lib = L.load()
    n, hw = 2, 16
    x = torch.zeros(n, hw, hw, 64, dtype=torch.bfloat16, device="cuda")
    out = torch.empty_like(x)
    w = torch.nn.Parameter(torch.zeros(64, 64, 3, 3, device="cuda"))
    pk = A.ops.pack_conv(w, None, torch.bfloat16)
    bits = torch.empty(n * hw * hw, 2, dtype=torch.int32, device="cuda")

    def args(**kw):
        a = L.ConvArgs(x=x.data_ptr(), x_pitch=64, x_coff=0, N=n, H=hw, W=hw, Cin=64, wpk=pk.wpk.data_ptr(), bias=0, CoutP=64, Cout=64, KH=3, KW=3,
                       relu=1, scale=1.0, res=0, mask=0, out=out.data_ptr(), out_pitch=64, out_coff=0, out_mode=L.OUT_NHWC, ps_r=0, post_add=0,
                       dtype=L.SRK_BF16, relu_bits=bits.data_ptr(), mask_bits=0)
        for k, v in kw.items():
            setattr(a, k, v)
        return a
    import ctypes as C
    assert lib.srk_conv_bits_ok(C.byref(args())) == 1
    assert lib.srk_conv_bits_ok(C.byref(args(ps_r=2, out_mode=L.OUT_NHWC_PS))) == 0
    assert lib.srk_conv_bits_ok(C.byref(args(ps_r=2))) == 0
    assert lib.srk_conv_bits_ok(C.byref(args(post_add=bits.data_ptr()))) == 0


def test_wdsr_b_finalizes_all_pointwise_pairs_with_one_launch(A, monkeypatch):
    """WDSR-B's 16 blocks issue 16 pw_wgrad_finalize launches.  The grouped form (opt-in: it measured no faster, the
    slabs are cache-hot right behind the kernel that wrote them) -- every block's srk_pw_wgrad_partial fills its slabs and ONE
    srk_pw_wgrad_finalize_group sums them all when the pass's deferred gradients are flushed -- gives the same bits as a finalize per pair."""
    L = A._lib
    kw = dict(type="B", n_feats=128, n_resblocks=3, scale_factor=2)

    def grads(each):
        prev = A.ops._PW_FIN_EACH
        A.ops._PW_FIN_EACH = each
        try:
            torch.manual_seed(0)
            m = A.WDSR(precision="bf16", **kw).cuda()
            g = torch.Generator().manual_seed(3)
            lr, hr = torch.rand(2, 3, 24, 24, generator=g).cuda(), torch.rand(2, 3, 48, 48, generator=g).cuda()
            loss = m._calculate_losses(img_sr=m(lr), img_hr=hr)["loss"]
            loss.backward()
            torch.cuda.synchronize()
            return {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
        finally:
            A.ops._PW_FIN_EACH = prev

    calls = []
    real, real_check = L.call, L.check
    monkeypatch.setattr(L, "call", lambda name, *a, **k: (calls.append(name), real(name, *a, **k))[1])
    monkeypatch.setattr(L, "check", lambda rc, name="": (calls.append(name), real_check(rc, name))[1])
    ga = grads(False)
    assert calls.count("srk_pw_wgrad_partial") == 3 and calls.count("srk_pw_wgrad_finalize_group") == 1 and calls.count("srk_pw_wgrad") == 0, \
        [c for c in calls if "pw_wgrad" in c]
    calls.clear()
    gb = grads(True)
    assert calls.count("srk_pw_wgrad") == 3 and calls.count("srk_pw_wgrad_partial") == 0
    assert ga.keys() == gb.keys() and len(ga) > 10
    for k in ga:
        assert torch.equal(ga[k], gb[k]), k


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(90, 3, 48, 48), (2, 3, 48, 48), (29, 3, 83, 85), (1, 3, 1, 5)])    # >= 200k pixels: the per-pixel kernel
def test_head_unfold_one_thread_per_pixel_is_exact(A, dtype, shape):
    """The 3-channel 3x3 im2col of the head conv (edsr.py:41-44 with sub_mean, common.py:58-71) has its own kernel (one thread per pixel, 64
    bytes per store group): the values are (x - mean) rounded once to the storage type, zeros outside the image and in channels 27..31, in
    torch's unfold order (channel-major, then kernel row, kernel column)."""
    torch.manual_seed(11)
    x = torch.rand(*shape, device="cuda")
    sub = torch.tensor([0.4488, 0.4371, 0.4040], device="cuda")
    got = A.ops.unfold_raw(x, sub, 3, dtype)
    n, c, h, w = shape
    ref = torch.nn.functional.unfold(x - sub.view(1, 3, 1, 1), 3, padding=1).view(n, 27, h, w).permute(0, 2, 3, 1).to(dtype)
    assert got.shape == (n, h, w, 32)
    assert torch.equal(got[..., :27], ref)
    assert not got[..., 27:].any()
    got0 = A.ops.unfold_raw(x, None, 3, dtype)
    ref0 = torch.nn.functional.unfold(x, 3, padding=1).view(n, 27, h, w).permute(0, 2, 3, 1).to(dtype)
    assert torch.equal(got0[..., :27], ref0)


_WG1X1_SCRIPT = r"""
import sys, torch
sys.path.insert(0, {root!r})
import sr_amd as A
torch.manual_seed(5)
out = {{}}
for tag, (n, h, w, cin, cout) in {{"head": (7, 48, 48, 32, 64), "ragged": (3, 13, 9, 32, 64), "one": (1, 1, 1, 32, 64), "sq": (2, 24, 24, 64, 64)}}.items():
    for dt in (torch.bfloat16, torch.float16):
        x = (torch.rand(n, h, w, cin, device="cuda") - 0.5).to(dt)
        x[..., 27:] = 0
        dy = (torch.rand(n, h, w, cout, device="cuda") - 0.5).to(dt)
        creal = 27 if cin == 32 else cin
        gw, gb = A.ops.wgrad_raw(x, dy, N=n, H=h, W=w, Cin=cin, Cout=cout, k=1, w_shape=(cout, creal, 1, 1), want_bias=True)
        ref = torch.einsum("nhwi,nhwo->oi", x.double(), dy.double())[:, :creal]
        out[f"{{tag}}_{{dt}}"] = (gw.cpu(), gb.cpu(), ref.cpu(), dy.double().sum((0, 1, 2)).cpu())
torch.save(out, sys.argv[1])
"""


def test_small_1x1_weight_gradient_kernel_matches_the_general_one_bit_for_bit(A, tmp_path):
    """The head conv's weight gradient (K = 27 (32) x 64) runs on a 64 x 64 kernel with a ring of 8 tile buffers instead of the 128 x 256
    one with two (conv_wgrad.hip): same tiles, same MFMA order -- the gradients must be the SAME BITS as the general kernel's (knob
    SRK_NO_WGRAD1X1_SMALL, read once per process: two processes), and right against float64."""
    import subprocess
    res = {}
    for knob in ("0", "1"):
        f = tmp_path / f"wg{knob}.pt"
        env = dict(os.environ, SRK_DEBUG="1")
        if knob == "1":
            env["SRK_NO_WGRAD1X1_SMALL"] = "1"
        else:
            env.pop("SRK_NO_WGRAD1X1_SMALL", None)
        out = subprocess.run([sys.executable, "-c", _WG1X1_SCRIPT.format(root=ROOT), str(f)], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout + out.stderr
        res[knob] = torch.load(f)
    assert res["0"].keys() == res["1"].keys() and len(res["0"]) == 8
    for k in res["0"]:
        gw0, gb0, ref, refb = res["0"][k]
        gw1, gb1, _, _ = res["1"][k]
        assert torch.equal(gw0, gw1) and torch.equal(gb0, gb1), k
        scale = ref.abs().max().item() + 1e-9
        assert (gw0.double().flatten(1) - ref).abs().max().item() <= 2e-5 * scale + 1e-6, k
        assert (gb0.double() - refb).abs().max().item() <= 2e-5 * (refb.abs().max().item() + 1.0), k


@pytest.mark.parametrize("name,kw", [("EDSR", dict(n_feats=64, n_resblocks=3, res_scale=0.1)), ("WDSR", dict(type="B", n_feats=32, n_resblocks=2))])
def test_graph_with_static_descriptor_tables_replays_like_eager(A, name, kw):
    """ops.graph_capture: the grouped launches' descriptor tables are written once at capture time instead of by upload launches inside the
    graph.  Replays on NEW data must give the gradients an eager pass gives on that data (the tables hold addresses only), also after other
    work has run between the replays (the tables' memory stays the graph's)."""
    torch.manual_seed(7)
    m = getattr(A.models, name)(scale_factor=2, channels=3, precision="bf16", patch_size=48, **kw).cuda()
    params = [p for p in m.parameters() if p.requires_grad]
    x = torch.rand(4, 3, 24, 24, device="cuda")
    y = torch.rand(4, 3, 48, 48, device="cuda")

    def fwd_bwd():
        for p in params:
            p.grad = None
        loss = A.ops.l1_loss(m(x), y)
        A.ops.backward(loss)
        return loss

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fwd_bwd()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with A.ops.graph_capture(g, stream=side) as holder:
        loss = fwd_bwd()
    assert holder.tables, "the backward pass of this model queues grouped launches: their tables must have gone through the holder"
    static_grads = [p.grad for p in params]
    for trial in range(3):
        x.copy_(torch.rand_like(x)); y.copy_(torch.rand_like(y))
        junk = [torch.rand(1 << 20, device="cuda") for _ in range(4)]         # other allocations and launches between the replays
        del junk
        g.replay()
        torch.cuda.synchronize()
        got = [t.clone() for t in static_grads]
        lg = loss.item()
        le = fwd_bwd().item()
        assert abs(lg - le) <= 1e-6 * max(1.0, abs(le))
        for a_, p in zip(got, params):
            assert torch.equal(a_, p.grad), (trial, tuple(a_.shape))
