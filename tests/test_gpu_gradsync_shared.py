"""Shared weights under GradSync; WDSR under the multi-rank graph step."""


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
