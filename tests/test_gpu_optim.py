"""GPU tests of sr_amd.optim.Adam (csrc/optim.hip: the Adam update of every parameter tensor in one launch) against
torch.optim.Adam, the optimizer the reference configures (models/srmodel.py:57-64,145-154).  Same update rule in fp32:
trajectories agree to a few ulp per step; state_dict layout, hipGraph replay, parameters without gradients, weight decay
and maximize follow torch."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(1,), (3,), (64,), (7, 5, 3, 3), (64, 64, 3, 3), (4097,), (33, 1000)]


@pytest.fixture(scope="module")
def A():
    import sr_amd
    assert torch.cuda.is_available()
    sr_amd._lib.load()
    return sr_amd


def _params(seed, dev="cuda"):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter((torch.rand(*s, generator=g) - 0.5).to(dev)) for s in SHAPES]


def _set_grads(ps, step, skip=()):
    g = torch.Generator().manual_seed(1000 + step)
    for i, p in enumerate(ps):
        gr = ((torch.rand(*p.shape, generator=g) - 0.5) * (10.0 ** (i % 4 - 2))).to(p.device)
        p.grad = None if i in skip else gr


@pytest.mark.parametrize("kw", [dict(), dict(lr=3e-2, betas=(0.8, 0.9), eps=1e-6), dict(weight_decay=0.1), dict(maximize=True)])
def test_matches_torch_adam(A, kw):
    ps, rs = _params(1), _params(1)
    opt, ropt = A.optim.Adam(ps, **kw), torch.optim.Adam(rs, **kw)
    assert isinstance(opt, torch.optim.Adam)
    for step in range(6):
        skip = (2,) if step in (1, 4) else ()               # a parameter without a gradient is left alone that step
        _set_grads(ps, step, skip)
        _set_grads(rs, step, skip)
        opt.step()
        ropt.step()
    torch.cuda.synchronize()
    for p, r in zip(ps, rs):
        assert float((p.detach() - r.detach()).abs().max()) <= 2e-6 * max(1.0, float(r.detach().abs().max())), p.shape
        st, rst = opt.state[p], ropt.state[r]
        assert float((st["exp_avg"] - rst["exp_avg"]).abs().max()) <= 1e-5 * float(rst["exp_avg"].abs().max()) + 1e-12
        assert float((st["exp_avg_sq"] - rst["exp_avg_sq"]).abs().max()) <= 1e-5 * float(rst["exp_avg_sq"].abs().max()) + 1e-20
    assert float(opt.state[ps[0]]["step"]) == 6.0


def test_state_dict_round_trip_and_torch_interchange(A):
    ps = _params(2)
    opt = A.optim.Adam(ps, lr=1e-2)
    for step in range(2):
        _set_grads(ps, step)
        opt.step()
    sd = copy.deepcopy(opt.state_dict())
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    # continue uninterrupted
    for step in range(2, 4):
        _set_grads(ps, step)
        opt.step()
    # resume from the checkpoint: into this optimizer and into torch's own Adam
    ps2 = _params(2)
    rs = _params(2)
    opt2, ropt = A.optim.Adam(ps2, lr=1e-2), torch.optim.Adam(rs, lr=1e-2)
    # parameters at checkpoint time: replay the first two steps
    tmp = A.optim.Adam(ps2, lr=1e-2)
    for step in range(2):
        _set_grads(ps2, step)
        tmp.step()
    with torch.no_grad():
        for r, p in zip(rs, ps2):
            r.copy_(p)
    opt2.load_state_dict(sd)
    ropt.load_state_dict(sd)
    for step in range(2, 4):
        _set_grads(ps2, step)
        _set_grads(rs, step)
        opt2.step()
        ropt.step()
    torch.cuda.synchronize()
    for p, q, r in zip(ps, ps2, rs):
        assert torch.equal(p, q), "resumed run == uninterrupted run, bit for bit"
        assert float((p.detach() - r.detach()).abs().max()) <= 2e-6 * max(1.0, float(r.detach().abs().max()))
    assert float(opt2.state[ps2[0]]["step"]) == 4.0


def test_graph_replay_advances_the_step(A):
    ps, rs = _params(3), _params(3)
    opt, ropt = A.optim.Adam(ps), torch.optim.Adam(rs)
    _set_grads(ps, 0)
    _set_grads(rs, 0)
    static = [p.grad for p in ps]
    opt.step()                                               # warm-up outside the capture (allocations, table upload)
    ropt.step()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            opt.step()
    for step in range(1, 5):
        gen = torch.Generator().manual_seed(500 + step)
        for sg, r in zip(static, rs):
            new = (torch.rand(*sg.shape, generator=gen) - 0.5).cuda()
            sg.copy_(new)
            r.grad = new.clone()
        g.replay()
        ropt.step()
    torch.cuda.synchronize()
    assert float(opt.state[ps[0]]["step"]) == 5.0
    for p, r in zip(ps, rs):
        assert float((p.detach() - r.detach()).abs().max()) <= 2e-6 * max(1.0, float(r.detach().abs().max()))


def test_eager_steps_between_replays_do_not_touch_the_captured_table(A):
    """a captured step uploads its (parameter, gradient) table through a memcpy node that re-reads the staging
    buffer on every replay.  An eager step with NEW gradient tensors between replays (trainer: the short last batch of an epoch)
    must not redirect the replays to its own gradients: capture, eager step with fresh gradients, >= 3 replays, compared
    tightly with torch.optim.Adam on the same gradient sequence."""
    ps, rs = _params(5), _params(5)
    opt, ropt = A.optim.Adam(ps), torch.optim.Adam(rs)
    _set_grads(ps, 0)
    _set_grads(rs, 0)
    static = [p.grad for p in ps]
    opt.step()
    ropt.step()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            opt.step()
    torch.cuda.current_stream().wait_stream(s)

    def fresh(step):
        gen = torch.Generator().manual_seed(900 + step)
        return [(torch.rand(*sg.shape, generator=gen) - 0.5).cuda() for sg in static]

    for step in range(1, 9):
        new = fresh(step)
        for r, n in zip(rs, new):
            r.grad = n.clone()
        if step in (1, 5):                                   # eager fallback: brand-new gradient tensors, p.grad re-pointed
            for p, n in zip(ps, new):
                p.grad = n.clone()
            opt.step()
            for p, sg in zip(ps, static):                    # what GraphedStep's graph sees again afterwards
                p.grad = sg
        else:
            for sg, n in zip(static, new):
                sg.copy_(n)
            g.replay()
        ropt.step()
    torch.cuda.synchronize()
    assert float(opt.state[ps[0]]["step"]) == 9.0
    for p, r in zip(ps, rs):
        assert float((p.detach() - r.detach()).abs().max()) <= 2e-6 * max(1.0, float(r.detach().abs().max()))
    plan = opt._plans[0]
    cap = plan.captured[-1]
    assert [g_.data_ptr() for g_ in cap.grads] == [sg.data_ptr() for sg in static], "the capture's table still names the static gradients"
    assert cap.table.data_ptr() != plan.eager.table.data_ptr() and cap.host.data_ptr() != plan.eager.host.data_ptr()


def test_errors(A):
    with pytest.raises(NotImplementedError):
        A.optim.Adam(_params(4), amsgrad=True)
    with pytest.raises(ValueError):
        A.optim.Adam(_params(4), lr=-1.0)
    ps = _params(4)
    opt = A.optim.Adam(ps)
    _set_grads(ps, 0)
    ps[6].grad = torch.rand(1000, 33, device="cuda").t()     # right shape, not contiguous
    with pytest.raises(RuntimeError, match="contiguous"):
        opt.step()
    opt = A.optim.Adam([torch.nn.Parameter(torch.zeros(3, device="cuda")), torch.nn.Parameter(torch.zeros(3))])
    with pytest.raises(RuntimeError, match="both CPU and GPU"):
        opt.step()


def test_model_trajectory_edsr_vs_torch_adam(A):
    """EDSR training through configure_optimizers(): 4 steps with the HIP Adam; a twin set of parameters is stepped by
    torch.optim.Adam on the SAME gradients (the model's own, including its tiny and zero entries) and must follow."""
    dev = torch.device("cuda")
    torch.manual_seed(0)
    m1 = A.EDSR(scale_factor=2, n_feats=32, n_resblocks=2, precision="bf16").to(dev)
    o1 = m1.configure_optimizers()[0]
    assert isinstance(o1, A.optim.Adam)
    ps = [p for p in m1.parameters() if p.requires_grad]
    twin = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    o2 = torch.optim.Adam(twin)
    g = torch.Generator().manual_seed(9)
    losses = []
    for step in range(4):
        lr = torch.rand(4, 3, 24, 24, generator=g).to(dev)
        hr = torch.rand(4, 3, 48, 48, generator=g).to(dev)
        o1.zero_grad(set_to_none=True)
        loss = m1._calculate_losses(img_sr=m1(lr), img_hr=hr)["loss"]
        loss.backward()
        losses.append(float(loss))
        for p, q in zip(ps, twin):
            q.grad = p.grad.detach().clone()
        o1.step()
        o2.step()
    torch.cuda.synchronize()
    for p, q in zip(ps, twin):
        assert float((p.detach() - q.detach()).abs().max()) <= 2e-6 * max(1.0, float(q.detach().abs().max()))
    assert losses[-1] < losses[0], "the model trains: forward sees the updated weights"
    assert all(p._version >= 4 for p in ps), "the update is visible to autograd's version counters"


def test_param_groups_lr_schedule_and_persistent_grads(A):
    """Two parameter groups with their own hyper-parameters, a learning rate changed between steps (what an lr scheduler does)
    and gradients that persist across steps (zero_grad(set_to_none=False)): still torch.optim.Adam's trajectory."""
    ps, rs = _params(7), _params(7)
    groups = lambda q: [dict(params=q[:3], lr=1e-2), dict(params=q[3:], lr=1e-3, betas=(0.5, 0.9), weight_decay=0.01)]
    opt, ropt = A.optim.Adam(groups(ps)), torch.optim.Adam(groups(rs))
    for step in range(5):
        _set_grads(ps, step)
        _set_grads(rs, step)
        if step == 0:
            kept = [p.grad for p in ps]
        else:                                                # same gradient tensors, new values (addresses unchanged: no table rebuild)
            for k, p in zip(kept, ps):
                k.copy_(p.grad)
                p.grad = k
        if step == 3:
            for o in (opt, ropt):
                o.param_groups[0]["lr"] = 5e-3
        opt.step()
        ropt.step()
        opt.zero_grad(set_to_none=False)
        assert all(float(p.grad.abs().max()) == 0.0 for p in ps)
    torch.cuda.synchronize()
    for p, r in zip(ps, rs):
        assert float((p.detach() - r.detach()).abs().max()) <= 3e-6 * max(1.0, float(r.detach().abs().max())), p.shape
