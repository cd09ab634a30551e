"""The device-resident loss scaler with several parameter groups."""


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
