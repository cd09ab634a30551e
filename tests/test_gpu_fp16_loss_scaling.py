"""fp16 training under hipGraph with device-resident dynamic loss scaling (BASELINE config 5's dtype; configs/all.yml:122), re-capture after a hyper-parameter change."""


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
