"""HrTailFn's parameter-gradient launches on a second stream (ops._hr_side_stream: small batches, beside the trunk's data-gradient chain): the gradients must be
the bits of the one-stream form, eagerly and in a replayed hipGraph, and the join must sit where the step first reads weight gradients."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def A():
    import sr_amd
    return sr_amd


def _grads(A, side, graph=False):
    ops = A.ops
    prev = ops._HR_SIDE
    ops._HR_SIDE = side
    used = []
    real = ops._hr_side_stream

    def spy(x, params):
        st = real(x, params)
        used.append(st is not None)
        return st
    ops._hr_side_stream = spy
    try:
        torch.manual_seed(0)
        m = A.EDSR(n_feats=64, n_resblocks=2, res_scale=0.1, scale_factor=4, precision="bf16").cuda()
        gen = torch.Generator().manual_seed(3)
        lr = torch.rand(4, 3, 24, 20, generator=gen).cuda()
        hr = torch.rand(4, 3, 96, 80, generator=gen).cuda()

        def step():
            for p in m.parameters():
                p.grad = None
            loss = m.training_step({"lr": lr, "hr": hr}, 0)["loss"]
            loss.backward()
            return loss
        loss = step()
        torch.cuda.synchronize()
        if graph:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                step()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with ops.graph_capture(g):
                loss = step()
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
        assert not ops._SIDE.pending, "every side-stream group is joined by the end of the backward pass"
        return float(loss), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}, used
    finally:
        ops._HR_SIDE = prev
        ops._hr_side_stream = real


@pytest.mark.parametrize("graph", [False, True])
def test_side_stream_gradients_are_the_one_stream_bits(A, graph):
    l0, g0, u0 = _grads(A, side=False, graph=graph)
    l1, g1, u1 = _grads(A, side=True, graph=graph)
    assert u1 and all(u1) and not any(u0), (u0, u1)
    assert l0 == l1
    assert set(g0) == set(g1) and len(g0) > 10
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k


def test_existing_grad_keeps_the_one_stream_form(A):
    """A parameter that already has a `.grad` is accumulated into by autograd right behind the node: no second stream then."""
    ops = A.ops
    torch.manual_seed(0)
    m = A.EDSR(n_feats=64, n_resblocks=1, res_scale=0.1, scale_factor=2, precision="bf16").cuda()
    lr, hr = torch.rand(2, 3, 16, 16).cuda(), torch.rand(2, 3, 32, 32).cuda()
    used = []
    real = ops._hr_side_stream

    def spy(x, params):
        st = real(x, params)
        used.append(st is not None)
        return st
    ops._hr_side_stream = spy
    try:
        m.training_step({"lr": lr, "hr": hr}, 0)["loss"].backward()
        first = {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}
        m.training_step({"lr": lr, "hr": hr}, 0)["loss"].backward()          # accumulates
        torch.cuda.synchronize()
    finally:
        ops._hr_side_stream = real
    assert used == [True, False], used
    for k, p in m.named_parameters():
        if p.grad is not None:
            assert torch.allclose(p.grad, 2 * first[k], rtol=1e-5, atol=1e-7), k
