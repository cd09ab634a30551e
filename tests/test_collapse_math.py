"""CPU check of the algebra behind the collapsed HR stage (tests/collapse_ref.py): the last upsampling stage + tail conv of EDSR /
RCAN / RDN (reference models/common.py:112-139, edsr.py:48-52) as one 5x5 convolution with border terms.  Everything is compared
with autograd of the reference's two-layer form in float64 -- including images so small that one pixel sits on two opposite edges."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import collapse_ref as R  # noqa: E402

DD = torch.float64


def _params(O, C, Ci, seed):
    g = torch.Generator().manual_seed(seed)
    mk = lambda *s: torch.randn(*s, dtype=DD, generator=g).requires_grad_(True)
    return mk(O, C, 3, 3), mk(O), mk(4 * C, Ci, 3, 3), mk(4 * C)


@pytest.mark.parametrize("hw", [(7, 9), (1, 1), (2, 3), (1, 5), (3, 1), (2, 2)])
@pytest.mark.parametrize("O", [3, 1])
def test_collapsed_forward_and_gradients_equal_the_two_layers(hw, O):
    H, W = hw
    Wt, bt, Wu, bu = _params(O, 5, 4, 11 + H * W + O)
    X = torch.randn(2, 4, H, W, dtype=DD, requires_grad=True)
    a = R.layerwise(X, Wt, bt, Wu, bu)
    b = R.collapsed_forward(X, Wt, bt, Wu, bu)
    assert a.shape == b.shape == (2, O, 2 * H, 2 * W)
    np.testing.assert_allclose(b.detach().numpy(), a.detach().numpy(), rtol=0, atol=1e-10 * float(a.abs().max()))
    g = torch.randn_like(a)
    ga = torch.autograd.grad(a, (X, Wt, bt, Wu, bu), g)
    gb = torch.autograd.grad(b, (X, Wt, bt, Wu, bu), g)
    for u, v in zip(ga, gb):
        np.testing.assert_allclose(v.numpy(), u.numpy(), rtol=0, atol=1e-10 * float(u.abs().max()))


@pytest.mark.parametrize("hw", [(4, 5), (1, 1), (2, 1), (1, 3)])
def test_explicit_chain_rule_formulas_equal_autograd(hw):
    """`np_border_sums` + `np_expand` (what srk_hrtail_edge_bwd_w / srk_hrtail_expand compute) against autograd of the two layers."""
    H, W = hw
    Wt, bt, Wu, bu = _params(2, 3, 2, 5 + H + W)
    X = torch.randn(2, 2, H, W, dtype=DD)
    y = R.layerwise(X, Wt, bt, Wu, bu)
    g = torch.randn_like(y)
    ref = torch.autograd.grad(y, (Wt, bt, Wu, bu), g)
    got = R.np_expand(*R.np_border_sums(X.numpy(), g.numpy()), Wt.detach().numpy(), Wu.detach().numpy(), bu.detach().numpy())
    for u, v in zip(ref, got):
        np.testing.assert_allclose(v, u.numpy(), rtol=0, atol=1e-10 * float(u.abs().max()))


def test_multiply_add_count():
    """The collapsed form needs 4 O Ci 25 multiply-adds per input pixel against 4 C Ci 9 + 4 O C 9 (EDSR-baseline: 7.7x fewer)."""
    O, C, Ci = 3, 64, 64
    assert (4 * C * Ci * 9 + 4 * O * C * 9) / (4 * O * Ci * 25) > 7.5
