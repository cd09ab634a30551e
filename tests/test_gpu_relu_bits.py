"""ReLU sign bits equal the activation mask (models/common.py:99-100 backward)."""


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
